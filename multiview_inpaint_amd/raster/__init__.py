"""Host-side mirror of the rasterizer plug-in the reference imports
(`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`,
gs-simp/gaussian_renderer/__init__.py:14), bound to the HIP C-ABI library with ctypes.

Call contract reproduced (gs-simp/gaussian_renderer/__init__.py:36-51, :85-93):
    settings = GaussianRasterizationSettings(image_height=..., image_width=..., tanfovx=..., tanfovy=...,
                                             bg=..., scale_modifier=..., viewmatrix=..., projmatrix=...,
                                             sh_degree=..., campos=..., prefiltered=...)      # 11 fields, no `debug`
    color, radii, depth = GaussianRasterizer(raster_settings=settings)(
        means3D=..., means2D=..., shs=..., colors_precomp=..., opacities=..., scales=..., rotations=...,
        cov3D_precomp=...)
backward returns gradients for means3D, means2D, shs|colors_precomp, opacities, scales+rotations|
cov3D_precomp; `means2D.grad[:, :2]` is what gs-simp/scene/gaussian_model.py:482-484 consumes.

PyTorch is used here only for device memory, the current stream and autograd plumbing.
"""
import ctypes as C
from typing import NamedTuple, Optional

import torch

from .. import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None or t.numel() == 0 else C.c_void_p(t.data_ptr())


def _dev_f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{what} must be on the GPU (got {t.device}); the rasterizer has no CPU path")
    return t.detach().to(torch.float32).contiguous()


class _Frame:
    """ctypes settings struct + the tensors it points into (kept alive together)."""

    def __init__(self, rs: GaussianRasterizationSettings):
        self.keep = [_dev_f32(rs.bg, "bg").reshape(-1), _dev_f32(rs.viewmatrix, "viewmatrix").reshape(-1),
                     _dev_f32(rs.projmatrix, "projmatrix").reshape(-1), _dev_f32(rs.campos, "campos").reshape(-1)]
        if self.keep[0].numel() != 3 or self.keep[1].numel() != 16 or self.keep[2].numel() != 16 or self.keep[3].numel() != 3:
            raise Exception("bg/campos must have 3 elements, viewmatrix/projmatrix 16")
        s = _lib.RasterSettings()
        s.image_height, s.image_width = int(rs.image_height), int(rs.image_width)
        s.tanfovx, s.tanfovy, s.scale_modifier = float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier)
        s.sh_degree, s.prefiltered = int(rs.sh_degree), int(bool(rs.prefiltered))
        s.bg, s.viewmatrix, s.projmatrix, s.campos = (t.data_ptr() for t in self.keep)
        self.c = s


class RasterState:
    """Scratch buffers one forward leaves behind for its backward (and for the parity tests)."""
    __slots__ = ("P", "M", "D", "W", "H", "geom", "binning", "image", "radii", "grad_rows", "rows_clean", "frame", "sources")

    def resolve_colors(self):
        """Evaluates the SH colours no tile needed (deferred colours, include/mvi_raster.h: mvi_raster_color_mode), so that
        the `rgbd` / `clamped` views are complete. Tests and introspection only; the forward's inputs are kept alive in
        `sources` for it."""
        if self.P == 0:
            return
        dev = self.geom.device
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mvi_raster_resolve_colors(C.byref(self.frame.c), self.P, _ptr(self.geom), self.geom.numel(), stream),
                       "resolve_colors")

    def take_rows(self, dev):
        """(accumulation rows [P,16], prezeroed flag) for one backward: the rows the forward zeroed inside its render kernel
        if it was asked to (`prepare_backward`) and nobody used them yet, else a fresh buffer the backward zeroes itself."""
        if getattr(self, "grad_rows", None) is not None and getattr(self, "rows_clean", False):
            self.rows_clean = False
            return self.grad_rows, 1
        return torch.empty(self.P, 16, dtype=torch.float32, device=dev), 0

    def views(self):
        L = _lib.lib()
        v = _lib.RasterViews()
        _lib.check(L.mvi_raster_get_views(self.P, self.D, self.W, self.H, _ptr(self.geom), _ptr(self.binning),
                                          _ptr(self.image), C.byref(v)), "get_views")
        return v

    def tensor(self, name, shape, dtype):
        """Copy of one intermediate array as a torch tensor (tests only)."""
        v = self.views()
        addr = getattr(v, name)
        n = 1
        for s in shape:
            n *= s
        if name == "tile_ids_sorted" and v.tile_id_bytes == 2 and torch.empty(0, dtype=dtype).element_size() == 4:
            raw = self.tensor(name, tuple(shape) + (2,), torch.uint8).to(torch.int32)       # little-endian uint16 words
            return (raw[..., 0] | (raw[..., 1] << 8)).to(dtype)
        out = torch.empty(shape, dtype=dtype, device=self.geom.device)
        if n:
            nbytes = n * out.element_size()
            for buf in (self.geom, self.binning, self.image):
                if buf is not None and buf.numel() and buf.data_ptr() <= addr < buf.data_ptr() + buf.numel():
                    off = addr - buf.data_ptr()
                    out.view(torch.uint8).reshape(-1).copy_(buf[off:off + nbytes])
                    return out
            raise RuntimeError(f"view {name} not inside a scratch buffer")
        return out


def _forward_render(L, fr, st, color, depth, stream, prepare_backward, dev):
    st.grad_rows, st.rows_clean = None, False
    if prepare_backward and st.P > 0:
        st.grad_rows = torch.empty(st.P, 16, dtype=torch.float32, device=dev)
        _lib.check(L.mvi_raster_forward_render_prepare(C.byref(fr.c), st.P, st.D, _ptr(st.radii), _ptr(st.geom), st.geom.numel(),
                                                       _ptr(st.binning), st.binning.numel(), _ptr(st.image), st.image.numel(),
                                                       _ptr(color), _ptr(depth), _ptr(st.grad_rows), stream),
                   "rasterize forward (render)")
        st.rows_clean = True
        return
    _lib.check(L.mvi_raster_forward_render(C.byref(fr.c), st.P, st.D, _ptr(st.radii), _ptr(st.geom), st.geom.numel(),
                                           _ptr(st.binning), st.binning.numel(), _ptr(st.image), st.image.numel(),
                                           _ptr(color), _ptr(depth), stream), "rasterize forward (render)")


def rasterize_forward(rs: GaussianRasterizationSettings, means3D, opacities, shs=None, colors_precomp=None,
                      scales=None, rotations=None, cov3D_precomp=None, prepare_backward=False):
    """Runs the HIP forward. Returns (color [3,H,W], radii [P] int32, depth [1,H,W], RasterState). prepare_backward: a
    backward will follow — its accumulation rows (64 B per Gaussian) are allocated now and zeroed inside the render kernel."""
    L = _lib.lib()
    fr = _Frame(rs)
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("means3D must be on the GPU; the rasterizer has no CPU path")
    P = int(means3D.shape[0])
    H, W = int(rs.image_height), int(rs.image_width)
    M = int(shs.shape[1]) if shs is not None and shs.numel() else 0
    st = RasterState()
    st.P, st.M, st.W, st.H = P, M, W, H
    st.frame, st.sources = fr, (means3D, shs)        # deferred SH colours read these until the render kernel has run
    u8 = dict(dtype=torch.uint8, device=dev)
    st.geom = torch.empty(L.mvi_raster_geom_bytes(P), **u8)
    st.image = torch.empty(L.mvi_raster_image_bytes(W, H), **u8)
    st.radii = torch.empty(P, dtype=torch.int32, device=dev)     # every entry is written by the preprocess kernel
    color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
    depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    D = C.c_int64(0)
    with torch.cuda.device(dev):
        _lib.check(L.mvi_raster_forward_geom(C.byref(fr.c), P, M, _ptr(means3D), _ptr(shs), _ptr(colors_precomp),
                                             _ptr(opacities), _ptr(scales), _ptr(rotations), _ptr(cov3D_precomp),
                                             _ptr(st.geom), st.geom.numel(), _ptr(st.radii), C.byref(D), stream),
                   "rasterize forward (geom)")
        st.D = int(D.value)
        st.binning = torch.empty(L.mvi_raster_binning_bytes(st.D, W, H) if st.D else 0, **u8)
        _forward_render(L, fr, st, color, depth, stream, prepare_backward, dev)
    return color, st.radii, depth, st


def rasterize_backward(rs: GaussianRasterizationSettings, st: RasterState, grad_color, means3D, shs=None,
                       colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, out=None, sh_grad="dense"):
    """Runs the HIP backward. Returns dict of gradients (None for inputs that were not given).
    `out` may hold preallocated contiguous fp32 tensors (e.g. dist.GradBucket.views) to write into.
    sh_grad (SH input only): "dense" -> g["shs"] [P,M,3]; "factor" -> g["sh_color_factor"] [P,3] instead, the
    colour factor of the rank-1 SH gradient (see sh_backward_views; view-parallel training exchanges this);
    "both" -> both."""
    if sh_grad not in ("dense", "factor", "both"):
        raise ValueError(f"sh_grad must be 'dense', 'factor' or 'both', got {sh_grad!r}")
    L = _lib.lib()
    fr = _Frame(rs)
    dev = means3D.device
    P = st.P
    f32 = dict(dtype=torch.float32, device=dev)
    out = out or {}

    def buf(name, *shape):
        t = out.get(name)
        if t is None:
            return torch.empty(*shape, **f32)
        if tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
            raise RuntimeError(f"out[{name!r}] must be a contiguous fp32 {shape} tensor on {dev}")
        return t
    g = dict(means3D=buf("means3D", P, 3), means2D=buf("means2D", P, 3), opacities=buf("opacities", P, 1),
             shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None)
    dcolors = None
    if shs is not None:
        if sh_grad != "factor":
            g["shs"] = buf("shs", P, st.M, 3)
        if sh_grad != "dense":
            dcolors = g["sh_color_factor"] = buf("sh_color_factor", P, 3)
    else:
        dcolors = g["colors_precomp"] = buf("colors_precomp", P, 3)
    if cov3D_precomp is not None:
        g["cov3D_precomp"] = buf("cov3D_precomp", P, 6)
    else:
        g["scales"], g["rotations"] = buf("scales", P, 3), buf("rotations", P, 4)
    scratch, clean = st.take_rows(dev)               # 64-byte accumulation row per Gaussian
    grad_color = grad_color.to(torch.float32).contiguous()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        _lib.check(L.mvi_raster_backward(
            C.byref(fr.c), P, st.M, st.D, _ptr(means3D), _ptr(shs), _ptr(colors_precomp), _ptr(scales), _ptr(rotations),
            _ptr(cov3D_precomp), _ptr(st.radii), _ptr(st.geom), _ptr(st.binning), _ptr(st.image), _ptr(grad_color),
            _ptr(g["means3D"]), _ptr(g["means2D"]), _ptr(g["opacities"]), _ptr(g["shs"]), _ptr(dcolors),
            _ptr(g["scales"]), _ptr(g["rotations"]), _ptr(g["cov3D_precomp"]), _ptr(scratch), clean, stream),
            "rasterize backward")
    return g


def rasterize_backward_split(rs: GaussianRasterizationSettings, st: RasterState, grad_color, means3D, shs, scales, rotations,
                             out, after_render=None):
    """rasterize_backward(..., sh_grad="factor") in two halves: after the render backward the colour factors are already
    in out["sh_color_factor"] and `after_render()` runs (dist.FactoredGradExchange starts its all-gather there), then the
    per-Gaussian chain rule fills the other gradients. Same kernels and results as the one-call form."""
    L = _lib.lib()
    fr = _Frame(rs)
    dev = means3D.device
    P = st.P
    f32 = dict(dtype=torch.float32, device=dev)
    g = {k: out[k] for k in ("means3D", "means2D", "opacities", "scales", "rotations", "sh_color_factor")}
    for k, shape in (("means3D", (P, 3)), ("means2D", (P, 3)), ("opacities", (P, 1)), ("scales", (P, 3)), ("rotations", (P, 4)),
                     ("sh_color_factor", (P, 3))):
        t = g[k]
        if tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
            raise RuntimeError(f"out[{k!r}] must be a contiguous fp32 {shape} tensor on {dev}")
    scratch, clean = st.take_rows(dev)
    spare = torch.empty(P, 3, **f32)                  # the chain-rule kernel writes the factors again: not into the buffer in flight
    grad_color = grad_color.to(torch.float32).contiguous()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        _lib.check(L.mvi_raster_backward_render(C.byref(fr.c), P, st.D, _ptr(st.radii), _ptr(st.geom), _ptr(st.binning),
                                                _ptr(st.image), _ptr(grad_color), _ptr(scratch), _ptr(g["sh_color_factor"]), 1,
                                                clean, stream), "rasterize backward (render)")
        if after_render is not None:
            after_render()
        _lib.check(L.mvi_raster_backward_geom(C.byref(fr.c), P, st.M, _ptr(means3D), _ptr(shs), None, _ptr(scales),
                                              _ptr(rotations), None, _ptr(st.radii), _ptr(st.geom), _ptr(scratch),
                                              _ptr(g["means3D"]), _ptr(g["means2D"]), _ptr(g["opacities"]), None, _ptr(spare),
                                              _ptr(g["scales"]), _ptr(g["rotations"]), None, stream),
                   "rasterize backward (geom)")
    return g


def rasterize_backward_ranged(rs: GaussianRasterizationSettings, st: RasterState, grad_color, means3D, shs, scales, rotations,
                              exchange, means2D_out=None):
    """Backward of one view for view-parallel training with the exchange overlapped range by range
    (dist.RangedGradExchange): render backward -> the colour factors' all-gather starts -> for every Gaussian range the
    chain rule (mvi_raster_backward_geom_range) writes straight into the exchange buffer and that range's all-reduce starts
    behind it -> exchange.finish() waits and rebuilds dL/dSH. Returns the summed gradients {means3D, shs, opacities,
    scales, rotations} and this view's dL/dmeans2D [P, 3] (a per-view densification statistic: not exchanged)."""
    L = _lib.lib()
    fr = _Frame(rs)
    dev = means3D.device
    P = st.P
    f32 = dict(dtype=torch.float32, device=dev)
    if exchange.P != P or exchange.M != st.M:
        raise RuntimeError(f"exchange was built for P={exchange.P}, M={exchange.M}; this view has P={P}, M={st.M}")
    factor = exchange.views["sh_color_factor"]
    m2d = torch.zeros(P, 3, **f32) if means2D_out is None else means2D_out
    scratch, clean = st.take_rows(dev)
    spare = torch.empty(P, 3, **f32)                  # the chain-rule kernel writes the factors again: not into the buffer in flight
    grad_color = grad_color.to(torch.float32).contiguous()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        _lib.check(L.mvi_raster_backward_render(C.byref(fr.c), P, st.D, _ptr(st.radii), _ptr(st.geom), _ptr(st.binning),
                                                _ptr(st.image), _ptr(grad_color), _ptr(scratch), _ptr(factor), 1, clean, stream),
                   "rasterize backward (render)")
        exchange.begin_gather(rs.campos)
        for r, (first, n) in enumerate(exchange.ranges):
            v = exchange.range_views(r)
            # the C entry takes the base of the FULL [P, w] array and moves to row `first` itself: hand it the address the
            # range's [n, w] block would have as rows first .. first + n of such an array
            base = lambda t, w: C.c_void_p(t.data_ptr() - 4 * w * first)
            _lib.check(L.mvi_raster_backward_geom_range(C.byref(fr.c), P, st.M, first, n, _ptr(means3D), _ptr(shs), None, _ptr(scales),
                                                        _ptr(rotations), None, _ptr(st.radii), _ptr(st.geom), _ptr(scratch),
                                                        base(v["means3D"], 3), _ptr(m2d), base(v["opacities"], 1), None, _ptr(spare),
                                                        base(v["scales"], 3), base(v["rotations"], 4), None, stream),
                       "rasterize backward (geom range)")
            exchange.reduce_range(r)
    g = exchange.finish(means3D)
    g["means2D"] = m2d
    return g


def sh_backward_views(means3D, campos, color_factors, M, sh_degree, out=None):
    """Sum over views of the SH gradient rebuilt from per-view colour factors:
    out[g,k,c] = sum_v Y_k(normalize(means3D[g] - campos[v])) * color_factors[v,g,c]  ->  [P,M,3].
    campos [V,3] and color_factors [V,P,3] may be strided views of one (all-gathered) buffer as long as their
    inner dimensions are contiguous."""
    L = _lib.lib()
    dev = means3D.device
    P = means3D.shape[0]
    V = color_factors.shape[0]
    if campos.shape != (V, 3) or color_factors.shape != (V, P, 3):
        raise ValueError(f"sh_backward_views: campos {tuple(campos.shape)} / color_factors {tuple(color_factors.shape)} "
                         f"do not match V={V}, P={P}")
    for t in (means3D, campos, color_factors):
        if t.dtype != torch.float32 or t.device != dev:
            raise RuntimeError("sh_backward_views: fp32 tensors on one device required")
    if not means3D.is_contiguous():
        means3D = means3D.contiguous()
    if V > 1 and (campos.stride(1) != 1 or color_factors.stride(2) != 1 or color_factors.stride(1) != 3):
        campos, color_factors = campos.contiguous(), color_factors.contiguous()
    elif V == 1:
        campos, color_factors = campos.contiguous(), color_factors.contiguous()
    if out is None:
        out = torch.empty(P, M, 3, dtype=torch.float32, device=dev)
    elif tuple(out.shape) != (P, M, 3) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
        raise RuntimeError(f"sh_backward_views: out must be a contiguous fp32 ({P}, {M}, 3) tensor on {dev}")
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        _lib.check(L.mvi_raster_sh_backward_views(P, M, int(sh_degree), V, _ptr(means3D), _ptr(campos),
                                                  campos.stride(0) if V > 1 else 3, _ptr(color_factors),
                                                  color_factors.stride(0) if V > 1 else 3 * P, _ptr(out), stream),
                   "sh_backward_views")
    return out


def rasterize_forward_raw(rs: GaussianRasterizationSettings, xyz, features_dc, features_rest, raw_opacity, raw_scaling,
                          raw_rotation, prepare_backward=False):
    """HIP forward fed with the GaussianModel's un-activated parameters (gaussian_model.py:95-115 happens in the
    kernels). Returns (color, radii, depth, RasterState) like rasterize_forward."""
    L = _lib.lib()
    fr = _Frame(rs)
    dev = xyz.device
    if dev.type != "cuda":
        raise RuntimeError("xyz must be on the GPU; the rasterizer has no CPU path")
    P, M = int(xyz.shape[0]), 1 + int(features_rest.shape[1])
    if features_dc.shape != (P, 1, 3) or features_rest.shape != (P, M - 1, 3) or raw_opacity.numel() != P or \
            raw_scaling.shape != (P, 3) or raw_rotation.shape != (P, 4):
        raise ValueError("rasterize_forward_raw: expected [P,3], [P,1,3], [P,M-1,3], [P,1], [P,3], [P,4]")
    H, W = int(rs.image_height), int(rs.image_width)
    st = RasterState()
    st.P, st.M, st.W, st.H = P, M, W, H
    st.frame, st.sources = fr, (xyz, features_dc, features_rest)
    u8 = dict(dtype=torch.uint8, device=dev)
    st.geom = torch.empty(L.mvi_raster_geom_bytes(P), **u8)
    st.image = torch.empty(L.mvi_raster_image_bytes(W, H), **u8)
    st.radii = torch.empty(P, dtype=torch.int32, device=dev)
    color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
    depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    D = C.c_int64(0)
    with torch.cuda.device(dev):
        _lib.check(L.mvi_raster_forward_geom_raw(C.byref(fr.c), P, M, _ptr(xyz), _ptr(features_dc), _ptr(features_rest),
                                                 _ptr(raw_opacity), _ptr(raw_scaling), _ptr(raw_rotation), _ptr(st.geom),
                                                 st.geom.numel(), _ptr(st.radii), C.byref(D), stream),
                   "rasterize forward raw (geom)")
        st.D = int(D.value)
        st.binning = torch.empty(L.mvi_raster_binning_bytes(st.D, W, H) if st.D else 0, **u8)
        _forward_render(L, fr, st, color, depth, stream, prepare_backward, dev)
    return color, st.radii, depth, st


def rasterize_backward_raw(rs: GaussianRasterizationSettings, st: RasterState, grad_color, xyz, features_dc, features_rest,
                           raw_opacity, raw_scaling, raw_rotation, out=None, sh_grad="dense"):
    """HIP backward with respect to the raw parameters. Returns dict(xyz, means2D, features_dc, features_rest,
    opacity, scaling, rotation). `out` may hold preallocated contiguous fp32 tensors to write into (same keys).
    sh_grad="factor" (view-parallel training, train_views.py): dict(xyz, means2D, opacity, scaling, rotation,
    sh_color_factor [P,3]) — the SH gradient stays in the factored form the exchange moves (dist.FactoredGradExchange) and
    features_dc / features_rest are None (mvi_raster_backward_raw_factor)."""
    if sh_grad not in ("dense", "factor"):
        raise ValueError(f"sh_grad must be 'dense' or 'factor', got {sh_grad!r}")
    L = _lib.lib()
    fr = _Frame(rs)
    dev, P, M = xyz.device, st.P, st.M
    f32 = dict(dtype=torch.float32, device=dev)
    out = out or {}

    def buf(name, *shape):
        t = out.get(name)
        if t is None:
            return torch.empty(*shape, **f32)
        if tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
            raise RuntimeError(f"out[{name!r}] must be a contiguous fp32 {shape} tensor on {dev}")
        return t
    g = dict(xyz=buf("xyz", P, 3), means2D=buf("means2D", P, 3), opacity=buf("opacity", P, 1), features_dc=None, features_rest=None,
             scaling=buf("scaling", P, 3), rotation=buf("rotation", P, 4))
    scratch, clean = st.take_rows(dev)
    grad_color = grad_color.to(torch.float32).contiguous()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        if sh_grad == "factor":
            g["sh_color_factor"] = buf("sh_color_factor", P, 3)
            _lib.check(L.mvi_raster_backward_raw_factor(
                C.byref(fr.c), P, M, st.D, _ptr(xyz), _ptr(features_dc), _ptr(features_rest), _ptr(raw_opacity), _ptr(raw_scaling),
                _ptr(raw_rotation), _ptr(st.radii), _ptr(st.geom), _ptr(st.binning), _ptr(st.image), _ptr(grad_color),
                _ptr(g["xyz"]), _ptr(g["means2D"]), _ptr(g["opacity"]), _ptr(g["sh_color_factor"]),
                _ptr(g["scaling"]), _ptr(g["rotation"]), _ptr(scratch), clean, stream), "rasterize backward raw (factor)")
            return g
        g["features_dc"], g["features_rest"] = buf("features_dc", P, 1, 3), buf("features_rest", P, M - 1, 3)
        _lib.check(L.mvi_raster_backward_raw(
            C.byref(fr.c), P, M, st.D, _ptr(xyz), _ptr(features_dc), _ptr(features_rest), _ptr(raw_opacity), _ptr(raw_scaling),
            _ptr(raw_rotation), _ptr(st.radii), _ptr(st.geom), _ptr(st.binning), _ptr(st.image), _ptr(grad_color),
            _ptr(g["xyz"]), _ptr(g["means2D"]), _ptr(g["opacity"]), _ptr(g["features_dc"]), _ptr(g["features_rest"]),
            _ptr(g["scaling"]), _ptr(g["rotation"]), _ptr(scratch), clean, stream), "rasterize backward raw")
    return g


class _RasterizeRaw(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, means2D, features_dc, features_rest, raw_opacity, raw_scaling, raw_rotation, rs):
        a = [t.detach().to(torch.float32).contiguous() for t in (xyz, features_dc, features_rest, raw_opacity, raw_scaling,
                                                                  raw_rotation)]
        need = any(ctx.needs_input_grad[:7])               # a backward will follow: zero its rows inside the render kernel
        color, radii, depth, st = rasterize_forward_raw(rs, *a, prepare_backward=need)
        ctx.rs, ctx.st, ctx.a = rs, st, a
        ctx.mark_non_differentiable(radii, depth)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_color, _gr, _gd):
        st = ctx.st
        if st.P == 0:
            return tuple(torch.zeros_like(t) for t in (ctx.a[0], ctx.a[0], *ctx.a[1:])) + (None,)
        g = rasterize_backward_raw(ctx.rs, st, grad_color, *ctx.a)
        return (g["xyz"], g["means2D"], g["features_dc"], g["features_rest"], g["opacity"].view_as(ctx.a[3]), g["scaling"],
                g["rotation"], None)


def _opt(t):
    """None or empty -> None; otherwise contiguous fp32."""
    if t is None or t.numel() == 0:
        return None
    return t.to(torch.float32).contiguous()


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs):
        a = dict(means3D=_opt(means3D), shs=_opt(shs), colors_precomp=_opt(colors_precomp), opacities=_opt(opacities),
                 scales=_opt(scales), rotations=_opt(rotations), cov3D_precomp=_opt(cov3D_precomp))
        if a["means3D"] is None:
            a["means3D"] = means3D.to(torch.float32).reshape(0, 3)
        color, radii, depth, st = rasterize_forward(rs, a["means3D"], a["opacities"], a["shs"], a["colors_precomp"],
                                                    a["scales"], a["rotations"], a["cov3D_precomp"],
                                                    prepare_backward=any(ctx.needs_input_grad[:8]))
        ctx.rs, ctx.st, ctx.a = rs, st, a
        ctx.mark_non_differentiable(radii, depth)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, _grad_depth):
        a, st = ctx.a, ctx.st
        if st.P == 0:
            z = lambda t: None if t is None else torch.zeros_like(t)
            return (z(a["means3D"]), None, z(a["shs"]), z(a["colors_precomp"]), z(a["opacities"]), z(a["scales"]),
                    z(a["rotations"]), z(a["cov3D_precomp"]), None)
        g = rasterize_backward(ctx.rs, st, grad_color, a["means3D"], a["shs"], a["colors_precomp"], a["scales"],
                               a["rotations"], a["cov3D_precomp"])
        return (g["means3D"], g["means2D"], g["shs"], g["colors_precomp"], g["opacities"], g["scales"],
                g["rotations"], g["cov3D_precomp"], None)


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """bool [P]: view-space z > 0.2 (part of the plug-in surface; no caller in the reference)."""
        L = _lib.lib()
        rs = self.raster_settings
        p = _dev_f32(positions, "positions")
        out = torch.zeros(p.shape[0], dtype=torch.uint8, device=p.device)
        fr = _Frame(rs)
        stream = C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)
        with torch.cuda.device(p.device):
            _lib.check(L.mvi_raster_mark_visible(p.shape[0], _ptr(p), _ptr(fr.keep[1]), _ptr(fr.keep[2]), _ptr(out), stream),
                       "mark_visible")
        return out.bool()

    def forward_raw(self, xyz, means2D, features_dc, features_rest, raw_opacity, raw_scaling, raw_rotation):
        """The same render fed with the GaussianModel's parameter tensors as stored (_xyz, _features_dc, _features_rest,
        _opacity, _scaling, _rotation): the activations of gaussian_model.py:95-115 and their chain rule run inside the
        preprocess kernels. Returns (color, radii, depth); gradients flow to the raw parameters and to means2D."""
        return _RasterizeRaw.apply(xyz, means2D, features_dc, features_rest, raw_opacity, raw_scaling, raw_rotation,
                                   self.raster_settings)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.empty(0, dtype=torch.float32, device=means3D.device)
        return _Rasterize.apply(means3D, means2D,
                                empty if shs is None else shs, empty if colors_precomp is None else colors_precomp,
                                opacities, empty if scales is None else scales, empty if rotations is None else rotations,
                                empty if cov3D_precomp is None else cov3D_precomp, self.raster_settings)
