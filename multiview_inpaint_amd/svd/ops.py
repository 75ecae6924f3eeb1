"""Device ops of the denoise loop. On a GPU tensor these run the hand-written HIP kernels through
the C-ABI library (include/mvi_unet_ops.h) and raise if it is missing — there is no silent
PyTorch substitute on the GPU inference path. On a CPU tensor they are plain PyTorch: that is
BASELINE.json configs[0] ("sgm VideoUNet single denoise step ... fp32 on CPU PyTorch"), the
reference's own CPU-runnable case, not a fallback for the GPU.

The HIP kernels are forward-only. Under autograd (ControlNet training — SURVEY.md §2 row 21, out of
scope) a GPU tensor that requires grad goes through PyTorch-ROCm's differentiable ops instead.

Reference ops: GroupNorm32 + SiLU (sgm/modules/diffusionmodules/util.py:259-276,
openaimodel.py:257-261,292-305), Normalize (sgm/modules/attention.py:125-128),
softmax(QK^T d^-1/2)V (sgm/modules/attention.py:332-336, :427-439).
"""
import os

import torch
import torch.nn.functional as F

# A GPU tensor leaves the HIP path for one of two reasons, treated differently:
#   * it requires grad (ControlNet training, out of scope): PyTorch-ROCm's differentiable ops run — the documented path;
#   * a shape / contiguity gate of a kernel fails under no_grad: that RAISES by default (STRICT_GATES; since round 3) — an
#     inference call never silently runs PyTorch ops in place of the kernels. MVI_STRICT=0 allows the substitute again
#     (recorded in FALLBACKS).
# Strict mode proper (MVI_STRICT=1, or ops.STRICT = True) raises in BOTH cases. The GPU tests and bench_svd.run_gpu run with
# it, so "the full-size step took the HIP branch everywhere" is asserted, not assumed.
STRICT = os.environ.get("MVI_STRICT", "") == "1"
STRICT_GATES = os.environ.get("MVI_STRICT", "") != "0"
FALLBACKS = []          # (op, reason) of every GPU-tensor fallback taken when not strict (diagnostics)


class HipPathError(RuntimeError):
    pass


def _fallback(t, op, reason):
    """Called right before a PyTorch substitute runs. CPU tensors: that IS the CPU path. GPU tensors: raise in strict
    mode, otherwise record."""
    if t is not None and t.is_cuda:
        if STRICT or (STRICT_GATES and reason != "requires grad"):
            raise HipPathError(f"{op}: GPU tensor left the HIP path ({reason}); set MVI_STRICT=0 to allow the PyTorch substitute")
        if len(FALLBACKS) < 4096:
            FALLBACKS.append((op, reason))


def _needs_autograd(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def _why(*ts):
    return "requires grad" if _needs_autograd(*ts) else "shape / layout gate"


def _stack3(y, T):
    """[(b T), C, ...] -> [(b T), 3C, ...]: frame t-1 | frame t | frame t+1 on the channel axis (zeros at the ends)."""
    bt, c = y.shape[:2]
    yv = y.reshape(bt // T, T, c, *y.shape[2:])
    out = y.new_zeros(bt // T, T, 3 * c, *y.shape[2:])
    out[:, 1:, :c] = yv[:, :-1]
    out[:, :, c:2 * c] = yv
    out[:, :-1, 2 * c:] = yv[:, 1:]
    return out.reshape(bt, 3 * c, *y.shape[2:])


def group_norm(x, num_groups, weight, bias, eps, silu=False, chan_bias=None):
    """GroupNorm over (C/G, *spatial) with fp32 statistics, optional fused SiLU; output dtype = x.dtype.
    chan_bias [N, C] (optional) is added to x first (the ResBlock's timestep-embedding bias)."""
    if x.is_cuda and not _needs_autograd(x, weight, bias, chan_bias):
        from . import hip_ops
        return hip_ops.group_norm_silu(x, num_groups, weight, bias, eps, silu, chan_bias=chan_bias)
    _fallback(x, "group_norm", _why(x, weight, bias, chan_bias))
    xf = x.float()
    if chan_bias is not None:
        xf = xf + chan_bias.float().reshape(*chan_bias.shape, *([1] * (x.ndim - 2)))
    y = F.group_norm(xf, num_groups, weight.float(), bias.float(), eps).type(x.dtype)
    return F.silu(y) if silu else y


def group_norm_tokens(x, num_groups, weight, bias, eps, silu=False, chan_bias=None):
    """group_norm(...) returned token-major: [N, C, *spatial] -> [N, prod(spatial), C] ("b c h w -> b (h w) c"
    fused into the normalisation's write)."""
    S = x[0, 0].numel()
    if x.is_cuda and not _needs_autograd(x, weight, bias, chan_bias):
        from . import hip_ops
        if x.shape[1] % 8 == 0 and S % 8 == 0:
            return hip_ops.group_norm_silu_tokens(x, num_groups, weight, bias, eps, silu, chan_bias=chan_bias)
        # odd channel / token counts: the plain HIP GroupNorm, then PyTorch's transpose copy (small tensors only)
        return hip_ops.group_norm_silu(x, num_groups, weight, bias, eps, silu, chan_bias=chan_bias).flatten(2).transpose(1, 2).contiguous()
    _fallback(x, "group_norm_tokens", _why(x, weight, bias, chan_bias))
    return group_norm(x, num_groups, weight, bias, eps, silu=silu, chan_bias=chan_bias).flatten(2).transpose(1, 2).contiguous()


def group_norm_tok2tok(t, num_groups, weight, bias, eps, silu=False, chan_bias=None, frames=1, partials=None):
    """GroupNorm(+SiLU) of token-major t [N, S, C] with token-major output (statistics per sample and group over (S, C/G));
    chan_bias [N, C] is added first. The norm between two convolutions that run on channels-last tensors. frames > 1: the temporal
    layers' norm — statistics over the `frames` consecutive samples of a video (video_model.py:71-75), chan_bias still per sample."""
    if t.is_cuda and not _needs_autograd(t, weight, bias, chan_bias):
        from . import hip_ops
        return hip_ops.group_norm_silu_tok2tok(t, num_groups, weight, bias, eps, silu, chan_bias=chan_bias, frames=frames, partials=partials)
    _fallback(t, "group_norm_tok2tok", _why(t, weight, bias, chan_bias))
    N, S, C = t.shape
    tf = t.float() if chan_bias is None else t.float() + chan_bias.float().reshape(N, 1, C)
    y = group_norm(tf.reshape(N // frames, frames * S, C).transpose(1, 2), num_groups, weight, bias, eps, silu=silu)
    return y.transpose(1, 2).reshape(N, S, C).to(t.dtype).contiguous()


def group_norm_frames(x, T, num_groups, weight, bias, eps, silu=False, chan_bias=None, stack3=False):
    """GroupNorm of the temporal layers — statistics over (C/G, T, H, W) per video — evaluated on the
    frame-major tensor x [(b T), C, H, W] the spatial layers produce (the reference permutes to
    b c t h w first: video_model.py:71-75). chan_bias [(b T), C] is added first; stack3 returns the
    result as [(b T), 3C, H, W] = (previous | own | next frame), the input of a (3,1,1) temporal
    convolution evaluated as one 1x1 convolution."""
    if x.is_cuda and not _needs_autograd(x, weight, bias, chan_bias):
        from . import hip_ops
        return hip_ops.group_norm_silu_frames(x, T, num_groups, weight, bias, eps, silu, chan_bias=chan_bias, stack3=stack3)
    _fallback(x, "group_norm_frames", _why(x, weight, bias, chan_bias))
    bt, c = x.shape[:2]
    xf = x.float()
    if chan_bias is not None:
        xf = xf + chan_bias.float().reshape(bt, c, *([1] * (x.ndim - 2)))
    x5 = xf.reshape(bt // T, T, c, *x.shape[2:]).transpose(1, 2)              # b c t h w
    y = F.group_norm(x5, num_groups, weight.float(), bias.float(), eps).type(x.dtype)
    if silu:
        y = F.silu(y)
    y = y.transpose(1, 2).reshape(x.shape)
    return _stack3(y, T) if stack3 else y


def attention(q, k, v, heads):
    """q [B,Sq,H*D], k/v [B,Sk,H*D] token-major as the Linear layers produce them -> [B,Sq,H*D]."""
    B, Sq, HD = q.shape
    Sk = k.shape[1]
    if q.is_cuda and not _needs_autograd(q, k, v):
        from . import hip_ops
        return hip_ops.attention(q, k, v, heads)
    _fallback(q, "attention", _why(q, k, v))
    D = HD // heads
    qh, kh, vh = (t.reshape(B, -1, heads, D).transpose(1, 2) for t in (q, k, v))
    o = F.scaled_dot_product_attention(qh, kh, vh)
    return o.transpose(1, 2).reshape(B, Sq, HD)


def packed_ok(x, heads, dim_head):
    """Whether the self-attention of x can take the packed-projection path (GPU inference, kernel-supported head dim)."""
    return x.is_cuda and not torch.is_grad_enabled() and dim_head in (16, 32, 64) and (heads * dim_head * x.element_size()) % 16 == 0


def attention_scale_fold_pays(x, dim_head):
    """Whether the self-attention of x [B, S, C] runs the 8-wave MFMA kernel (csrc/attn_flash8.hip), whose softmax is bound by
    vector issue: the only kernel for which a q that carries the softmax scale (CrossAttention._packed_qkv_weight(fold=True))
    is faster."""
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.dim() == 3):
        return False
    from . import hip_ops
    return hip_ops.attention_kernel_variant(x.shape[1], x.shape[1], dim_head, x.dtype) in (8, 16)


def _unfold_q(q, heads, q_log2):
    """q of a packed projection whose q rows carry dim_head^-1/2 * log2(e) (CrossAttention._packed_qkv_weight), for a consumer that
    applies the softmax scale itself."""
    if not q_log2:
        return q
    D = q.shape[-1] // heads
    return (q.float() * (D ** 0.5 * 0.6931471805599453)).to(q.dtype)


def attention_packed(qkv, heads, q_log2=False):
    """Self-attention on one packed projection qkv [B, S, 3*H*D] = (q | k | v) -> [B, S, H*D]. q_log2: the q third already
    carries dim_head^-1/2 * log2(e) (folded into its projection's weights)."""
    if qkv.is_cuda and not _needs_autograd(qkv):
        from . import hip_ops
        return hip_ops.attention_packed(qkv.contiguous(), heads, q_log2=q_log2)
    _fallback(qkv, "attention_packed", _why(qkv))
    q, k, v = qkv.chunk(3, dim=-1)
    return attention(_unfold_q(q, heads, q_log2).contiguous(), k.contiguous(), v.contiguous(), heads)


def attention_temporal_packed(qkv, heads, T, q_log2=False):
    """attention_temporal on a packed projection [(bo*T), S, 3*H*D]."""
    if qkv.is_cuda and not _needs_autograd(qkv):
        from . import hip_ops
        return hip_ops.attention_temporal_packed(qkv.contiguous(), heads, T, q_log2=q_log2)
    _fallback(qkv, "attention_temporal_packed", _why(qkv))
    q, k, v = qkv.chunk(3, dim=-1)
    return attention_temporal(_unfold_q(q, heads, q_log2).contiguous(), k.contiguous(), v.contiguous(), heads, T)


def attention_wide(q, k, v):
    """Single-head attention whose head is the whole channel axis (model.py:180-195: D = C = 512): q [B,Sq,D],
    k/v [B,Sk,D] -> [B,Sq,D]."""
    if q.is_cuda and not _needs_autograd(q, k, v):
        from . import hip_ops
        if q.shape[-1] <= 64 and q.shape[-1] in (16, 32, 64):
            return hip_ops.attention(q, k, v, 1)
        return hip_ops.attention_wide(q.contiguous(), k.contiguous(), v.contiguous())
    _fallback(q, "attention_wide", _why(q, k, v))
    return F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]


def attention_temporal(q, k, v, heads, T):
    """Self-attention over the frame axis without regrouping tokens: q/k/v [(bo*T), S, H*D] ->
    same shape; one softmax per (video, spatial token, head) over its T frames. Equals
    `(b t) s c -> (b s) t c`, attention, and the inverse regrouping of the reference
    (sgm/modules/video_attention.py:115, :136-140)."""
    BT, S, HD = q.shape
    if q.is_cuda and not _needs_autograd(q, k, v):
        from . import hip_ops
        return hip_ops.attention_temporal(q, k, v, heads, T)
    _fallback(q, "attention_temporal", _why(q, k, v))
    bo = BT // T

    def regroup(t):
        return t.reshape(bo, T, S, HD).transpose(1, 2).reshape(bo * S, T, HD)
    o = attention(regroup(q), regroup(k), regroup(v), heads)
    return o.reshape(bo, S, T, HD).transpose(1, 2).reshape(BT, S, HD)


def geglu(h):
    """h [..., 2*inner] -> h[..., :inner] * gelu(h[..., inner:]) (sgm/modules/attention.py:93-95)."""
    inner = h.shape[-1] // 2
    if h.is_cuda and not _needs_autograd(h) and inner % 8 == 0:
        from . import hip_ops
        return hip_ops.geglu(h)
    _fallback(h, "geglu", _why(h))
    a, gate = h.chunk(2, dim=-1)
    return a * F.gelu(gate)


FF_GEGLU_MIN_ROWS = 32768       # below, the fused kernel's 256-row blocks do not fill the chip: library GEMM + geglu
K320_KERNELS = os.environ.get("MVI_K320", "1") != "0"      # MVI_K320=0: library GEMMs everywhere (same-box A/B runs)


FF_GEGLU_N320 = os.environ.get("MVI_FF_GEGLU_N320", "1") != "0"
FF_GEGLU_N320_K = (640, 1280)
FF_GEGLU_N320_MIN_ROWS = int(os.environ.get("MVI_FF_GEGLU_N320_MIN_ROWS", "16000"))     # (level 2: 16 128 rows; round 5, with the 16x16x32 kernel: 385 us against 329 + 84)


def linear_geglu(x, weight, bias=None):
    """GEGLU of the reference (sgm/modules/attention.py:87-95): `x, gate = F.linear(x, weight, bias).chunk(2, -1); x * F.gelu(gate)`.
    On the GPU, for the shapes csrc/ff_geglu.hip covers (K = 320 in bf16 / f16: the level-0 FeedForward layers) and enough rows,
    projection and gating run as ONE kernel and the [rows, 2 inner] intermediate never exists; everything else is the library
    GEMM followed by geglu()."""
    if K320_KERNELS and x.is_cuda and not _needs_autograd(x, weight, bias) and x.dtype == weight.dtype:
        from . import hip_ops
        rows = x.numel() // max(x.shape[-1], 1)
        if rows >= FF_GEGLU_MIN_ROWS and hip_ops.ff_geglu_supported(x.shape[-1], weight.shape[0] // 2, x.dtype):
            return hip_ops.ff_geglu(x, weight, bias)
        # K = 640 / 1280 (levels 1 and 2): csrc/linear_n320.hip's GEGLU form, where it was measured faster than the library
        # GEMM + geglu_kernel (FF_GEGLU_N320_MIN_ROWS; MVI_FF_GEGLU_N320=0: library everywhere)
        if FF_GEGLU_N320 and x.shape[-1] in FF_GEGLU_N320_K and rows >= FF_GEGLU_N320_MIN_ROWS \
                and hip_ops.ff_geglu_n320_supported(x.shape[-1], weight.shape[0] // 2, x.dtype):
            return hip_ops.ff_geglu_n320(x, weight, bias)
    return geglu(F.linear(x, weight, bias))


N320_KERNEL = os.environ.get("MVI_N320", "1") != "0"          # csrc/linear_n320.hip for [rows, K] x [K, 320] with rows >= FF_GEGLU_MIN_ROWS
# Round 5: the same kernel with 320 g outputs (g column groups per row block) for the projections of levels 1 and 2 INTO 640 / 1280
# channels — FeedForward.net[2] (K = 2560 / 5120) and the attention output projections (K = 640 / 1280) — where it was measured faster
# than the tuned library GEMM (180 / 165 / 55 / 46 us against 215 / 189 / 80 / 57: profiles/round5_n320_groups.txt); the packed q/k/v
# projections (1920 / 3840 outputs) tie and stay with the library, and so does everything whose grid would not fill the chip (level 3).
N320_GROUPS = os.environ.get("MVI_N320_GROUPS", "1") != "0"
N320_GROUP_WIDTHS = (640, 1280)
N320_GROUP_MIN_BLOCKS = 200


def linear(x, weight, bias=None):
    """F.linear(x, weight, bias). On the GPU, the K = 320 projections of the level-0 transformer blocks (packed q/k/v, to_out,
    proj_in / proj_out: output-bound GEMMs around a 20-step loop) take csrc/ff_geglu.hip's plain-epilogue kernel
    (mvi_linear_k320), the level-0 projections INTO 320 channels with a long contraction (FeedForward.net[2], K = 1280) take
    csrc/linear_n320.hip (mvi_linear_n320), and so do the projections into 640 / 1280 channels of levels 1 and 2 (N320_GROUPS above);
    everything else is the library GEMM."""
    if K320_KERNELS and N320_KERNEL and N320_GROUPS and x.is_cuda and weight.shape[0] in N320_GROUP_WIDTHS and x.dtype == weight.dtype \
            and not _needs_autograd(x, weight, bias):
        rows = x.numel() // max(x.shape[-1], 1)
        if (rows + 255) // 256 * (weight.shape[0] // 320) >= N320_GROUP_MIN_BLOCKS:
            from . import hip_ops
            if hip_ops.linear_n320_supported(x.shape[-1], weight.shape[0], x.dtype):
                return hip_ops.linear_n320(x, weight, bias)
    if K320_KERNELS and x.is_cuda and not _needs_autograd(x, weight, bias) and x.dtype == weight.dtype \
            and x.numel() // max(x.shape[-1], 1) >= FF_GEGLU_MIN_ROWS:
        from . import hip_ops
        n320 = N320_KERNEL and weight.shape[0] == 320 and hip_ops.linear_n320_supported(x.shape[-1], weight.shape[0], x.dtype)
        if n320 and x.shape[-1] == 320:            # 320 -> 320 (to_out, proj_in / proj_out): both kernels apply, this one is 6 % faster (93 / 99 us)
            return hip_ops.linear_n320(x, weight, bias)
        if hip_ops.linear_k320_supported(x.shape[-1], weight.shape[0], x.dtype):
            return hip_ops.linear_k320(x, weight, bias)
        if n320:
            return hip_ops.linear_n320(x, weight, bias)
    return F.linear(x, weight, bias)


def linear_module(mod, x):
    """`mod(x)` for an nn.Linear (or Sequential(Linear, Dropout) at inference) through linear()."""
    if isinstance(mod, torch.nn.Sequential) and len(mod) == 2 and isinstance(mod[0], torch.nn.Linear) \
            and isinstance(mod[1], torch.nn.Dropout) and not (mod[1].training and mod[1].p > 0):
        return linear(x, mod[0].weight, mod[0].bias)
    if type(mod) is torch.nn.Linear:
        return linear(x, mod.weight, mod.bias)
    return mod(x)


def bias_residual_add(h, bias=None, x=None):
    """h [N, C, *spatial] + bias[c] + x in one pass (conv bias and ResBlock skip add, openaimodel.py:354)."""
    if h.is_cuda and not _needs_autograd(h, bias, x):
        from . import hip_ops
        return hip_ops.bias_residual_add(h, bias, x)
    _fallback(h, "bias_residual_add", _why(h, bias, x))
    out = h
    if bias is not None:
        out = out + bias.to(h.dtype).reshape(1, -1, *([1] * (h.ndim - 2)))
    if x is not None:
        out = out + x
    return out


def concat_add(h, skip, ctrl=None):
    """torch.cat([h, skip + ctrl], dim=1) — the decoder's skip concatenation with the ControlNet residual added on the
    way (models/csvd.py:79-91) — as one pass over contiguous NCHW activations."""
    if (h.is_cuda and not _needs_autograd(h, skip, ctrl) and h.dtype == skip.dtype and (ctrl is None or ctrl.dtype == h.dtype)
            and h.is_contiguous() and skip.is_contiguous() and (ctrl is None or (ctrl.is_contiguous() and ctrl.shape == skip.shape))
            and h.shape[0] < 65536):
        from . import hip_ops
        return hip_ops.concat_add(h, skip, ctrl)
    _fallback(h, "concat_add", _why(h, skip, ctrl))
    return torch.cat([h, skip if ctrl is None else skip + ctrl], dim=1)


def bias_residual_blend(h, bias, x, alpha):
    """alpha * x + (1 - alpha) * (x + h + bias[c]) = x + (1 - alpha) * (h + bias[c]) in one pass; alpha [N] per sample
    (the temporal ResBlock's skip add followed by AlphaBlender, video_model.py:67-81, util.py:358-372)."""
    if h.is_cuda and not _needs_autograd(h, bias, x, alpha):
        from . import hip_ops
        return hip_ops.bias_residual_blend(h, bias, x, alpha)
    _fallback(h, "bias_residual_blend", _why(h, bias, x, alpha))
    xt = bias_residual_add(h, bias, x)
    return torch.lerp(xt, x, alpha.reshape(-1, *([1] * (h.ndim - 1))).to(x.dtype))


def bias_silu(h, bias):
    """silu(h + bias[c]) for a convolution output h [N, C, *spatial] whose bias was withheld (one pass; may reuse h)."""
    if h.is_cuda and not _needs_autograd(h, bias) and h.is_contiguous():
        from . import hip_ops
        return hip_ops.bias_silu(h, bias)
    _fallback(h, "bias_silu", _why(h, bias))
    if bias is not None:
        h = h + bias.to(h.dtype).reshape(1, -1, *([1] * (h.ndim - 2)))
    return F.silu(h)


def add_layer_norm(x, norm, h=None, row=None, ret_pre=False):
    """Residual add(s) fused with the next LayerNorm: s_pre = x + h, s = s_pre + row, y = norm(s) for token-major
    x [B, S, C]; `row` [G, 1, C] (G divides B*S) is broadcast over equal runs of rows — the single-token
    cross-attention row or the frame-index embedding. Returns (y, s, s_pre); s is x when h and row are None,
    s_pre is returned only with ret_pre (and is s when row is None). `norm` is the nn.LayerNorm."""
    C_ = x.shape[-1]
    if x.is_cuda and not _needs_autograd(x, h, row, norm.weight, norm.bias):
        from . import hip_ops
        if norm.elementwise_affine and hip_ops.layernorm_supported(C_, x.dtype):
            y, s, s_pre = hip_ops.add_layer_norm(x, norm.weight, norm.bias, norm.eps, h=h, row=row, ret_pre=ret_pre)
            return y, (x if s is None else s), s_pre
    _fallback(x, "add_layer_norm", _why(x, h, row, norm.weight, norm.bias))
    s_pre = x if h is None else x + h
    s = s_pre
    if row is not None:
        G = row.reshape(-1, C_).shape[0]
        rows = x.numel() // C_
        s = (s_pre.reshape(G, rows // G, C_) + row.reshape(G, 1, C_)).reshape(x.shape)
    return norm(s), s, (s_pre if ret_pre else None)


LN_EPILOGUE = os.environ.get("MVI_LN_EPILOGUE", "1") != "0"    # MVI_LN_EPILOGUE=0: projection and add + LayerNorm as two kernels (same-box A/B)


def _plain_linear(mod):
    if isinstance(mod, torch.nn.Sequential) and len(mod) == 2 and isinstance(mod[0], torch.nn.Linear) \
            and isinstance(mod[1], torch.nn.Dropout) and not (mod[1].training and mod[1].p > 0):
        return mod[0]
    return mod if type(mod) is torch.nn.Linear else None


def linear_add_layer_norm(lin_in, lin_mod, resid, norm, row=None, ret_pre=False):
    """add_layer_norm(resid, norm, h=lin_mod(lin_in), row=row, ret_pre=ret_pre) — the projection that ends an attention / FeedForward
    layer, the residual add(s) behind it and the LayerNorm of the NEXT layer (attention.py:544-572, video_attention.py:110-141).
    With resid None the sum starts from the projection itself (proj_in followed by norm1). On the GPU at the level-0 width (320
    outputs: a block of csrc/linear_n320.hip holds whole rows) all of it is ONE kernel: the projection's result and its read-back
    never touch memory (hip_ops.linear_n320_add_layer_norm); everywhere else it is linear_module + add_layer_norm."""
    lin = _plain_linear(lin_mod)
    if LN_EPILOGUE and K320_KERNELS and N320_KERNEL and lin is not None and lin_in.is_cuda and lin.out_features == 320 \
            and isinstance(norm, torch.nn.LayerNorm) and norm.elementwise_affine and tuple(norm.normalized_shape) == (320,) \
            and lin_in.dtype == lin.weight.dtype and lin_in.numel() // max(lin_in.shape[-1], 1) >= FF_GEGLU_MIN_ROWS \
            and not _needs_autograd(lin_in, lin.weight, lin.bias, resid, row, norm.weight, norm.bias):
        from . import hip_ops
        if hip_ops.linear_n320_supported(lin_in.shape[-1], 320, lin_in.dtype) and (resid is None or resid.dtype == lin_in.dtype) \
                and (row is None or row.dtype == lin_in.dtype):
            return hip_ops.linear_n320_add_layer_norm(lin_in, lin.weight, lin.bias, norm.weight, norm.bias, norm.eps, resid=resid, row=row,
                                                      ret_pre=ret_pre)
    return finish_add_layer_norm(linear_module(lin_mod, lin_in), dict(resid=resid, norm=norm, row=row, ret_pre=ret_pre))


def finish_add_layer_norm(h, fuse):
    """The unfused form of a `fuse` request ({resid, norm, row, ret_pre}) for a layer output h that already exists."""
    resid, row, ret_pre = fuse.get("resid"), fuse.get("row"), fuse.get("ret_pre", False)
    if resid is None:
        return add_layer_norm(h, fuse["norm"], row=row, ret_pre=ret_pre)
    return add_layer_norm(resid, fuse["norm"], h=h, row=row, ret_pre=ret_pre)


def add_lerp(x, h, base, alpha):
    """lerp(x + h, base, alpha): alpha * base + (1 - alpha) * (x + h) with alpha [G] broadcast over equal runs of
    the rows of x [B, S, C] (AlphaBlender after the temporal block's last residual add)."""
    if x.is_cuda and not _needs_autograd(x, h, base, alpha) and x.shape[-1] % 8 == 0:
        from . import hip_ops
        return hip_ops.add_lerp(x, h, base, alpha)
    _fallback(x, "add_lerp", _why(x, h, base, alpha))
    t = x if h is None else x + h
    C_ = x.shape[-1]
    G = alpha.numel()
    a = alpha.reshape(G, 1, 1).to(x.dtype)
    return torch.lerp(t.reshape(G, -1, C_), base.reshape(G, -1, C_), a).reshape(x.shape)


def tokens_to_planes_add(tok, x_in, bias=None):
    """tok [B, (h w), C] (+ bias[c]) -> [B, C, h, w] plus x_in, one pass (SpatialTransformer's exit; ResBlock's exit when its
    convolutions ran channels-last)."""
    if tok.is_cuda and not _needs_autograd(tok, x_in, bias):
        from . import hip_ops
        if tok.shape[-1] % 8 == 0 and tok.shape[1] % 8 == 0:
            return hip_ops.tokens_to_planes_add(tok, x_in, bias)
    _fallback(tok, "tokens_to_planes_add", _why(tok, x_in))
    t = tok if bias is None else tok + bias.to(tok.dtype)
    return t.transpose(1, 2).reshape(x_in.shape) + x_in
