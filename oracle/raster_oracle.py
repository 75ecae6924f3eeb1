"""ctypes front-end of oracle/raster_oracle.c — the CPU restatement of the rasterizer-with-depth.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
PARITY UNPINNED for the third-party CUDA extension it restates — see the header of
raster_oracle.c for what the reference tree does pin and how that is checked.

Boundary restated: gs-simp/gaussian_renderer/__init__.py:36-51, :85-93.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libraster_oracle.so")
TILE = 16


class Params(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("sh_degree", C.c_int32), ("M", C.c_int32),
        ("W", C.c_int32), ("H", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("viewmatrix", C.c_float * 16), ("projmatrix", C.c_float * 16),
        ("campos", C.c_float * 3), ("bg", C.c_float * 3),
        ("prefiltered", C.c_int32),
    ]


def build(force=False):
    src = os.path.join(_HERE, "raster_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_scan.restype = C.c_int64
        _lib.orc_key_bits.restype = C.c_int
        _lib.orc_set_threads.argtypes = [C.c_int]
        _lib.orc_get_threads.restype = C.c_int
    return _lib


def set_threads(n):
    """Host threads of the per-Gaussian / per-pixel loops (bench.py's cpu_baseline only; the tests keep 1, see raster_oracle.c)."""
    lib().orc_set_threads(int(n))


def _p(a, ty=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def make_params(P, sh_degree, M, W, H, tanfovx, tanfovy, scale_modifier, viewmatrix, projmatrix,
                campos, bg):
    p = Params()
    p.P, p.sh_degree, p.M, p.W, p.H = int(P), int(sh_degree), int(M), int(W), int(H)
    p.tanfovx, p.tanfovy, p.scale_modifier = float(tanfovx), float(tanfovy), float(scale_modifier)
    p.viewmatrix[:] = [float(v) for v in np.asarray(viewmatrix, np.float32).reshape(-1)]
    p.projmatrix[:] = [float(v) for v in np.asarray(projmatrix, np.float32).reshape(-1)]
    p.campos[:] = [float(v) for v in np.asarray(campos, np.float32).reshape(-1)]
    p.bg[:] = [float(v) for v in np.asarray(bg, np.float32).reshape(-1)]
    p.prefiltered = 0
    return p


def forward(params, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
            cov3D_precomp=None, render=True):
    """Full forward. Returns a dict holding every intermediate the parity tests compare."""
    L = lib()
    P, W, H = params.P, params.W, params.H
    means3D, opacities = _f32(means3D), _f32(opacities).reshape(-1)
    shs, colors_precomp = _f32(shs), _f32(colors_precomp)
    scales, rotations, cov3D_precomp = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    o = dict(
        radii=np.zeros(P, np.int32), xy=np.zeros((P, 2), np.float32), depths=np.zeros(P, np.float32),
        cov3D=np.zeros((P, 6), np.float32), rgb=np.zeros((P, 3), np.float32),
        conic_opacity=np.zeros((P, 4), np.float32), tiles_touched=np.zeros(P, np.uint32),
        clamped=np.zeros((P, 3), np.uint8),
    )
    L.orc_preprocess_forward(C.byref(params), _p(means3D), _p(scales), _p(rotations), _p(opacities),
                             _p(shs), _p(colors_precomp), _p(cov3D_precomp), _p(o["radii"]), _p(o["xy"]),
                             _p(o["depths"]), _p(o["cov3D"]), _p(o["rgb"]), _p(o["conic_opacity"]),
                             _p(o["tiles_touched"]), _p(o["clamped"]))
    o["offsets"] = np.zeros(P, np.uint32)
    D = int(L.orc_scan(C.c_int32(P), _p(o["tiles_touched"]), _p(o["offsets"])))
    o["num_rendered"] = D
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    keys = np.zeros(max(D, 1), np.uint64)
    vals = np.zeros(max(D, 1), np.uint32)
    L.orc_duplicate_with_keys(C.byref(params), _p(o["xy"]), _p(o["depths"]), _p(o["radii"]),
                              _p(o["offsets"]), _p(keys), _p(vals))
    o["keys_unsorted"], o["values_unsorted"] = keys[:D], vals[:D]
    bits = int(L.orc_key_bits(C.c_int(gx * gy)))
    ks = np.zeros(max(D, 1), np.uint64)
    vs = np.zeros(max(D, 1), np.uint32)
    L.orc_sort_pairs(C.c_int64(D), C.c_int(bits), _p(keys), _p(vals), _p(ks), _p(vs))
    o["keys_sorted"], o["point_list"] = ks[:D], vs[:D]
    ranges = np.zeros((gx * gy, 2), np.uint32)
    L.orc_tile_ranges(C.c_int64(D), _p(ks), C.c_int32(gx * gy), _p(ranges))
    o["ranges"] = ranges
    if render:
        o["color"] = np.zeros((3, H, W), np.float32)
        o["depth"] = np.zeros((1, H, W), np.float32)
        o["final_T"] = np.zeros((H, W), np.float32)
        o["n_contrib"] = np.zeros((H, W), np.uint32)
        L.orc_render_forward(C.byref(params), _p(ranges), _p(vs), _p(o["xy"]), _p(o["rgb"]),
                             _p(o["conic_opacity"]), _p(o["depths"]), _p(o["color"]), _p(o["depth"]),
                             _p(o["final_T"]), _p(o["n_contrib"]))
    return o


def margins(params, fwd, rel_margin=2e-5):
    """Per-pixel bitmask [H, W] u8 of decisions taken within `rel_margin` of their threshold (orc_render_margins): bit 0
    colour / final_T / n_contrib, bit 1 median depth. rel_margin covers the 1-2 ulp of an exp() plus the rounding that a
    product of a few hundred (1 - alpha) factors accumulates in T."""
    L = lib()
    out = np.zeros((params.H, params.W), np.uint8)
    vs = np.ascontiguousarray(fwd["point_list"]) if fwd["num_rendered"] else np.zeros(1, np.uint32)
    L.orc_render_margins(C.byref(params), _p(fwd["ranges"]), _p(vs), _p(fwd["xy"]), _p(fwd["conic_opacity"]),
                         C.c_float(rel_margin), _p(out))
    return out


def backward(params, fwd, dL_dcolor_img, means3D, shs=None, colors_precomp=None, scales=None,
             rotations=None, cov3D_precomp=None):
    """Analytic backward. `fwd` is the dict returned by forward(). Returns the gradients the
    autograd Function hands back: means3D, means2D ([P,3], z column 0), shs | colors_precomp,
    opacities ([P,1]), scales, rotations | cov3D_precomp."""
    L = lib()
    P = params.P
    means3D = _f32(means3D)
    shs, colors_precomp = _f32(shs), _f32(colors_precomp)
    scales, rotations, cov3D_precomp = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    g = _f32(dL_dcolor_img)
    d2 = np.zeros((P, 2), np.float32)
    dcon = np.zeros((P, 3), np.float32)
    dop = np.zeros(P, np.float32)
    dcol = np.zeros((P, 3), np.float32)
    vs = np.ascontiguousarray(fwd["point_list"]) if fwd["num_rendered"] else np.zeros(1, np.uint32)
    L.orc_render_backward(C.byref(params), _p(fwd["ranges"]), _p(vs), _p(fwd["xy"]), _p(fwd["rgb"]),
                          _p(fwd["conic_opacity"]), _p(fwd["final_T"]), _p(fwd["n_contrib"]), _p(g),
                          _p(d2), _p(dcon), _p(dop), _p(dcol))
    dmeans = np.zeros((P, 3), np.float32)
    dshs = None if shs is None else np.zeros((P, params.M, 3), np.float32)
    dcov = None if cov3D_precomp is None else np.zeros((P, 6), np.float32)
    dsc = None if cov3D_precomp is not None else np.zeros((P, 3), np.float32)
    drot = None if cov3D_precomp is not None else np.zeros((P, 4), np.float32)
    L.orc_preprocess_backward(C.byref(params), _p(means3D), _p(scales), _p(rotations), _p(shs),
                              _p(cov3D_precomp), _p(fwd["radii"]), _p(fwd["cov3D"]), _p(fwd["clamped"]),
                              _p(d2), _p(dcon), _p(dcol), _p(dmeans), _p(dshs), _p(dcov), _p(dsc), _p(drot))
    means2D = np.zeros((P, 3), np.float32)
    means2D[:, :2] = d2
    return dict(means3D=dmeans, means2D=means2D, shs=dshs,
                colors_precomp=(dcol if colors_precomp is not None else None),
                opacities=dop.reshape(P, 1), scales=dsc, rotations=drot, cov3D_precomp=dcov,
                dL_dconic=dcon, dL_dcolor=dcol)
