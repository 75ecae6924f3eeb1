"""ctypes binding of the C-ABI HIP library (include/mvi_raster.h, include/mvi_unet_ops.h, include/mvi_train_ops.h).

There is NO fallback: if libmvi_hip.so is missing or does not export a declared symbol this
module raises, and every op of the package fails with it."""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVI_HIP_LIB: another build of the SAME library for same-box A/B timing (tools/ab_lib.sh); never set by tests or bench defaults
LIB_PATH = os.environ.get("MVI_HIP_LIB") or os.path.join(_HERE, "csrc", "libmvi_hip.so")
LIB_OVERRIDDEN = bool(os.environ.get("MVI_HIP_LIB"))
if LIB_OVERRIDDEN:                                       # an experiment build may compute wrong results on purpose: never silently
    import sys as _sys
    _sys.stderr.write(f"[mvi] MVI_HIP_LIB is set: every kernel comes from {LIB_PATH}, not from the in-tree build\n")
INCLUDE_DIR = os.path.normpath(os.path.join(_HERE, "..", "include"))


class RasterSettings(C.Structure):
    """struct mvi_raster_settings (include/mvi_raster.h)"""
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("sh_degree", C.c_int32), ("prefiltered", C.c_int32),
        ("bg", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
    ]


class RasterViews(C.Structure):
    """struct mvi_raster_views"""
    _fields_ = [(n, C.c_void_p) for n in (
        "depths", "means2D", "cov3D_a", "cov3D_b", "conic_opacity", "rgbd", "tiles_touched", "clamped",
        "tile_ids_sorted", "point_list", "ranges", "final_T", "n_contrib")] + [("tile_id_bytes", C.c_int32),
                                                                                   ("grad_support", C.c_void_p)]


class AdamGroup(C.Structure):
    """struct mvi_adam_group (include/mvi_train_ops.h)"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_float)]


class CompactTensor(C.Structure):
    """struct mvi_compact_tensor (include/mvi_train_ops.h)"""
    _fields_ = [("in_", C.c_void_p), ("out", C.c_void_p), ("width", C.c_int32), ("packed_stride", C.c_int32)]


def declared_symbols():
    """Every function name declared in include/*.h."""
    names = []
    for fn in sorted(os.listdir(INCLUDE_DIR)):
        if fn.endswith(".h"):
            src = open(os.path.join(INCLUDE_DIR, fn)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names += re.findall(r"\b(mvi_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def raster_source_digest():
    """sha1 over the rasterizer kernel sources (csrc/raster_*.hip, raster_common.h): the identity of the build that a
    committed counter file (profiles/raster_traffic.json) was measured on. Sources, not the .so: hipcc's output is not
    byte-reproducible, and .git does not travel to the GPU box."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(_HERE, "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.startswith("raster_") and fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -m multiview_inpaint_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback for this path.")
    L = C.CDLL(LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    if missing:
        raise RuntimeError(f"{LIB_PATH} does not export {missing}; rebuild it")
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    L.mvi_raster_last_error.restype = C.c_char_p
    L.mvi_version.restype = C.c_char_p
    L.mvi_raster_geom_bytes.restype = sz
    L.mvi_raster_geom_bytes.argtypes = [i32]
    L.mvi_raster_image_bytes.restype = sz
    L.mvi_raster_image_bytes.argtypes = [i32, i32]
    L.mvi_raster_binning_bytes.restype = sz
    L.mvi_raster_binning_bytes.argtypes = [i64, i32, i32]
    L.mvi_raster_forward_geom.restype = C.c_int
    L.mvi_raster_forward_geom.argtypes = [C.POINTER(RasterSettings), i32, i32] + [vp] * 7 + [vp, sz, vp, C.POINTER(i64), vp]
    L.mvi_raster_forward_render.restype = C.c_int
    L.mvi_raster_forward_render.argtypes = [C.POINTER(RasterSettings), i32, i64, vp, vp, sz, vp, sz, vp, sz, vp, vp, vp]
    L.mvi_raster_forward_render_prepare.restype = C.c_int
    L.mvi_raster_forward_render_prepare.argtypes = [C.POINTER(RasterSettings), i32, i64, vp, vp, sz, vp, sz, vp, sz, vp, vp, vp, vp]
    L.mvi_raster_backward.restype = C.c_int
    L.mvi_raster_backward.argtypes = [C.POINTER(RasterSettings), i32, i32, i64] + [vp] * 20 + [i32, vp]
    L.mvi_raster_backward_render.restype = C.c_int
    L.mvi_raster_backward_render.argtypes = [C.POINTER(RasterSettings), i32, i64] + [vp] * 7 + [i32, i32, vp]
    L.mvi_raster_backward_geom.restype = C.c_int
    L.mvi_raster_backward_geom.argtypes = [C.POINTER(RasterSettings), i32, i32] + [vp] * 18
    L.mvi_raster_backward_geom_range.restype = C.c_int
    L.mvi_raster_backward_geom_range.argtypes = [C.POINTER(RasterSettings), i32, i32, i32, i32] + [vp] * 18
    L.mvi_raster_forward_geom_raw.restype = C.c_int
    L.mvi_raster_forward_geom_raw.argtypes = [C.POINTER(RasterSettings), i32, i32] + [vp] * 6 + [vp, sz, vp, C.POINTER(i64), vp]
    L.mvi_raster_backward_raw.restype = C.c_int
    L.mvi_raster_backward_raw.argtypes = [C.POINTER(RasterSettings), i32, i32, i64] + [vp] * 19 + [i32, vp]
    L.mvi_raster_backward_raw_factor.restype = C.c_int
    L.mvi_raster_backward_raw_factor.argtypes = [C.POINTER(RasterSettings), i32, i32, i64] + [vp] * 18 + [i32, vp]
    L.mvi_raster_sh_backward_views.restype = C.c_int
    L.mvi_raster_sh_backward_views.argtypes = [i32, i32, i32, i32, vp, vp, i64, vp, i64, vp, vp]
    L.mvi_raster_mark_visible.restype = C.c_int
    L.mvi_raster_mark_visible.argtypes = [i32, vp, vp, vp, vp, vp]
    L.mvi_raster_get_views.restype = C.c_int
    L.mvi_raster_get_views.argtypes = [i32, i64, i32, i32, vp, vp, vp, C.POINTER(RasterViews)]
    L.mvi_raster_timing_enable.restype = C.c_int
    L.mvi_raster_timing_enable.argtypes = [C.c_int]
    L.mvi_raster_timing_enable_stages.restype = C.c_int
    L.mvi_raster_timing_enable_stages.argtypes = [C.c_uint32]
    L.mvi_raster_timing_read.restype = C.c_int
    L.mvi_raster_timing_read.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    L.mvi_raster_stage_name.restype = C.c_char_p
    L.mvi_raster_stage_name.argtypes = [C.c_int]
    L.mvi_raster_binning_version.restype = C.c_int
    L.mvi_raster_binning_version.argtypes = [C.c_int]
    L.mvi_raster_backward_mode.restype = C.c_int
    L.mvi_raster_backward_mode.argtypes = [C.c_int]
    L.mvi_raster_color_mode.restype = C.c_int
    L.mvi_raster_color_mode.argtypes = [C.c_int]
    L.mvi_raster_resolve_colors.restype = C.c_int
    L.mvi_raster_resolve_colors.argtypes = [C.POINTER(RasterSettings), i32, vp, sz, vp]
    L.mvi_raster_dev_stamps.restype = C.c_int
    L.mvi_raster_dev_stamps.argtypes = [C.c_int, vp]
    _bind_unet_ops(L)
    _bind_train_ops(L)
    _lib = L
    return L


def _bind_unet_ops(L):
    # filled in by the UNet-ops section once include/mvi_unet_ops.h exists
    try:
        from . import _unet_ops_bind
    except ImportError:
        return
    _unet_ops_bind.bind(L)


def _bind_train_ops(L):
    """include/mvi_train_ops.h"""
    vp, i32, sz, f32 = C.c_void_p, C.c_int32, C.c_size_t, C.c_float
    L.mvi_photometric_loss_workspace_bytes.restype = sz
    L.mvi_photometric_loss_workspace_bytes.argtypes = [i32, i32]
    L.mvi_photometric_loss.restype = C.c_int
    L.mvi_photometric_loss.argtypes = [vp, vp, vp, i32, i32, f32, f32, vp, vp, vp, sz, vp]
    L.mvi_photometric_loss_stats.restype = C.c_int
    L.mvi_photometric_loss_stats.argtypes = [vp, vp, vp, i32, i32, vp, vp, sz, vp]
    L.mvi_photometric_loss_grad2.restype = C.c_int
    L.mvi_photometric_loss_grad2.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, sz, vp]
    L.mvi_train_last_error.restype = C.c_char_p
    L.mvi_knn3_mean_dist2.restype = C.c_int
    L.mvi_knn3_mean_dist2.argtypes = [vp, i32, vp, vp]
    L.mvi_adam_step.restype = C.c_int
    L.mvi_adam_step.argtypes = [C.POINTER(AdamGroup), i32, C.c_double, C.c_double, C.c_double, i32, vp]
    L.mvi_compact_workspace_bytes.restype = sz
    L.mvi_compact_workspace_bytes.argtypes = [i32]
    L.mvi_compact_plan.restype = C.c_int
    L.mvi_compact_plan.argtypes = [vp, i32, vp, sz, vp, vp]
    L.mvi_compact_gather.restype = C.c_int
    L.mvi_compact_gather.argtypes = [C.POINTER(CompactTensor), i32, i32, C.c_uint32, vp, vp]
    L.mvi_support_pack_bits.restype = C.c_int
    L.mvi_support_pack_bits.argtypes = [vp, i32, vp, vp]
    L.mvi_support_union_bits.restype = C.c_int
    L.mvi_support_union_bits.argtypes = [vp, i32, i32, vp, vp]
    for f in (L.mvi_compact_gather_window, L.mvi_compact_scatter_window):
        f.restype = C.c_int
        f.argtypes = [C.POINTER(CompactTensor), i32, i32, vp, C.c_uint32, C.c_uint32, vp, vp]
    L.mvi_gaussian_activations.restype = C.c_int
    L.mvi_gaussian_activations.argtypes = [i32, i32] + [vp] * 10
    L.mvi_gaussian_activations_backward.restype = C.c_int
    L.mvi_gaussian_activations_backward.argtypes = [i32, i32] + [vp] * 13


def check(rc, what):
    if rc != 0:
        msg = lib().mvi_raster_last_error().decode(errors="replace")
        if rc == -1:
            raise Exception(msg)            # argument-combination errors: plain Exception like the plug-in
        raise RuntimeError(f"{what} failed ({rc}): {msg}")
