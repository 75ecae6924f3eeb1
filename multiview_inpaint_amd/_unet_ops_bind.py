"""ctypes signatures of include/mvi_unet_ops.h (bound by _lib.lib())."""
import ctypes as C


def bind(L):
    vp, i32, i64, sz, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_float
    L.mvi_groupnorm_workspace_bytes.restype = sz
    L.mvi_groupnorm_workspace_bytes.argtypes = [i64, i32, i64, i32]
    L.mvi_groupnorm_silu.restype = C.c_int
    L.mvi_groupnorm_silu.argtypes = [vp, vp, vp, vp, i64, i32, i64, i32, f32, i32, i32, vp, sz, vp]
    L.mvi_groupnorm_silu_temporal.restype = C.c_int
    L.mvi_groupnorm_silu_temporal.argtypes = [vp, vp, vp, vp, i64, i32, i32, i64, i32, f32, i32, i32, vp, sz, vp]
    L.mvi_groupnorm_silu_ex.restype = C.c_int
    L.mvi_groupnorm_silu_ex.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i64, i32, f32, i32, i32, i32, vp, sz, vp]
    L.mvi_groupnorm_sync_bytes.restype = sz
    L.mvi_groupnorm_sync_bytes.argtypes = []
    L.mvi_groupnorm_silu_ex2.restype = C.c_int
    L.mvi_groupnorm_silu_ex2.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i64, i32, f32, i32, i32, i32, vp, sz, vp, sz, vp]
    L.mvi_groupnorm_silu_tokens.restype = C.c_int
    L.mvi_groupnorm_silu_tokens.argtypes = [vp, vp, vp, vp, vp, i64, i32, i64, i32, f32, i32, i32, vp, sz, vp]
    L.mvi_attention_forward.restype = C.c_int
    L.mvi_attention_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp]
    L.mvi_attention_temporal.restype = C.c_int
    L.mvi_attention_temporal.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp]
    L.mvi_attention_forward_strided.restype = C.c_int
    L.mvi_attention_forward_strided.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, i64, i64, i64, vp]
    L.mvi_attention_temporal_strided.restype = C.c_int
    L.mvi_attention_temporal_strided.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, i64, i64, vp]
    L.mvi_bias_silu.restype = C.c_int
    L.mvi_bias_silu.argtypes = [vp, vp, vp, i64, i32, i64, i32, vp]
    L.mvi_geglu.restype = C.c_int
    L.mvi_geglu.argtypes = [vp, vp, i64, i32, i32, vp]
    L.mvi_ff_geglu_supported.restype = C.c_int
    L.mvi_ff_geglu_supported.argtypes = [i32, i32, i32]
    L.mvi_ff_geglu.restype = C.c_int
    L.mvi_ff_geglu_out_rows.restype = i64
    L.mvi_ff_geglu_out_rows.argtypes = [i64]
    L.mvi_ff_geglu.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, i64, i64, i32, vp]
    L.mvi_concat_add.restype = C.c_int
    L.mvi_concat_add.argtypes = [vp, vp, vp, vp, i64, i32, i32, i64, i32, vp]
    L.mvi_bias_residual_blend.restype = C.c_int
    L.mvi_bias_residual_blend.argtypes = [vp, vp, vp, vp, vp, i64, i32, i64, i32, vp]
    L.mvi_bias_residual_add.restype = C.c_int
    L.mvi_bias_residual_add.argtypes = [vp, vp, vp, vp, i64, i32, i64, i32, vp]
    L.mvi_add_layernorm.restype = C.c_int
    L.mvi_add_layernorm.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp]
    L.mvi_layernorm_supported.restype = C.c_int
    L.mvi_layernorm_supported.argtypes = [i32, i32]
    L.mvi_add_lerp.restype = C.c_int
    L.mvi_add_lerp.argtypes = [vp, vp, vp, vp, i64, vp, i64, i32, i32, vp]
    L.mvi_tokens_to_planes_add.restype = C.c_int
    L.mvi_tokens_to_planes_add.argtypes = [vp, vp, vp, i64, i32, i64, i32, vp]
    L.mvi_softmax_rows.restype = C.c_int
    L.mvi_softmax_rows.argtypes = [vp, i64, i32, f32, i32, vp]
    L.mvi_attention_kernel_kind.restype = C.c_int
    L.mvi_attention_kernel_kind.argtypes = [i32, i32, i32, i32]
    L.mvi_unet_last_error.restype = C.c_char_p
