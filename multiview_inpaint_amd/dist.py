"""View-parallel multi-GPU reconstruction: one process per GPU, Gaussian parameters
replicated, every rank rasterizes a different camera view, ONE all-reduce (RCCL over xGMI; gloo in
the CPU tests) of a flat gradient bucket per step, plus the small side-channel reductions that keep
densification (gs-simp/scene/gaussian_model.py:466-484, gs-simp/train.py:113-116) identical on
every rank. The reference itself has no distributed 3DGS code (process per scene,
gs-simp/train.sh:1); its single-GPU loop is gs-simp/inpaint_rec.py:71-172.

No collective runs inside the rasterizer: the exchange step is the gradient sum only.

Wire volume. The plain exchange all-reduces [3 | 3M | 1 | 3 | 4] floats per Gaussian (354 MB at M = 16,
P = 1.5 M); over the point-to-point xGMI links a ring all-reduce moves 2 (W-1)/W of that per GPU. The SH part
(3M of the 11 + 3M floats) is rank-1 per view and Gaussian — dL/dSH[k][c] = Y_k(view direction) * g[c] — so
FactoredGradExchange all-gathers the 3-float colour factor g of every view (+ the camera centres) and rebuilds
the summed SH gradient locally (mvi_raster_sh_backward_views): (W-1) * 12 B + all-reduce of 11 floats per
Gaussian instead of an all-reduce of 59, i.e. 241 MB instead of 620 MB in and out of each GPU at W = 8, M = 16.
"""
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as td


def shard_views(views: Sequence, rank: int, world: int) -> List:
    """Rank r takes views r, r+world, r+2*world, ... of the (identically ordered) camera list."""
    return [v for i, v in enumerate(views) if i % world == rank]


class GradBucket:
    """One contiguous fp32 buffer holding the rasterizer's gradient outputs as SoA segments
    [means3D 3 | shs 3M | opacities 1 | scales 3 | rotations 4] x P, so the per-step exchange is a
    single large all-reduce (84 / 138 / 354 MB at sh_degree 0 / 1 / 3 for P = 1.5 M). means2D (the
    densification statistic, summed as a per-view norm, not as a vector) lives outside the bucket."""

    SEGMENTS = (("means3D", 3), ("shs", None), ("opacities", 1), ("scales", 3), ("rotations", 4))

    def __init__(self, P: int, M: int, device, group=None):
        self.P, self.M, self.group = P, M, group
        widths = [(n, (3 * M if w is None else w)) for n, w in self.SEGMENTS]
        self.flat = torch.zeros(P * sum(w for _, w in widths), dtype=torch.float32, device=device)
        self.views: Dict[str, Optional[torch.Tensor]] = {}
        o = 0
        for n, w in widths:
            v = self.flat[o:o + P * w]
            self.views[n] = v.view(P, M, 3) if n == "shs" else v.view(P, w)
            o += P * w
        self.views["means2D"] = torch.zeros(P, 3, dtype=torch.float32, device=device)

    def all_reduce(self, async_op: bool = False):
        return td.all_reduce(self.flat, op=td.ReduceOp.SUM, group=self.group, async_op=async_op)


def all_reduce_param_grads(params: Iterable[torch.Tensor], group=None, average: bool = False):
    """Sums `.grad` of the Gaussian parameter tensors (the reference's six groups: _xyz, _features_dc,
    _features_rest, _opacity, _scaling, _rotation — gaussian_model.py:154-163) across ranks through
    one flat bucket. Parameters without a grad contribute zeros so every rank sends the same shape."""
    params = [p for p in params]
    if not params:
        return
    dev = params[0].device
    sizes = [p.numel() for p in params]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    o = 0
    for p, n in zip(params, sizes):
        if p.grad is not None:
            flat[o:o + n].copy_(p.grad.reshape(-1))
        o += n
    td.all_reduce(flat, op=td.ReduceOp.SUM, group=group)
    if average:
        flat /= td.get_world_size(group)
    o = 0
    for p, n in zip(params, sizes):
        g = flat[o:o + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        o += n


def reduce_densification_stats(viewspace_grad: torch.Tensor, visibility: torch.Tensor, radii: torch.Tensor,
                               xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor,
                               group=None, norm_scale: float = 1.0):
    """Multi-view counterpart of add_densification_stats (gaussian_model.py:482-484) + the max-radii
    update (train.py:115): each rank contributes the norm of ITS view's screen-space gradient where
    ITS view saw the Gaussian; sums / max are taken over ranks so every rank holds identical stats.
    norm_scale undoes a scaling of the loss the caller applied for the optimizer's sake (reduce="mean": upstream = 1 / world):
    the statistic the reference thresholds (densify_grad_threshold, gaussian_model.py:467-470) is the per-view norm of the
    gradient of the UNSCALED loss, averaged over the views that saw the Gaussian."""
    # mask-free (no nonzero() read-back): where(mask, norm, 0) over the whole arrays — the same values as the indexed form
    vis = visibility.view_as(xyz_gradient_accum)
    norm = torch.where(vis, torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True).to(xyz_gradient_accum.dtype), 0.0)
    if norm_scale != 1.0:
        norm = norm * norm_scale
    cnt = visibility.to(denom.dtype).view_as(denom)
    rad = torch.where(visibility, radii.to(max_radii2D.dtype), torch.zeros_like(max_radii2D))
    pack = torch.cat([norm.reshape(-1), cnt.reshape(-1)])
    td.all_reduce(pack, op=td.ReduceOp.SUM, group=group)
    td.all_reduce(rad, op=td.ReduceOp.MAX, group=group)
    n = norm.numel()
    xyz_gradient_accum += pack[:n].view_as(xyz_gradient_accum)
    denom += pack[n:].view_as(denom)
    torch.maximum(max_radii2D, rad, out=max_radii2D)


_SH_C0 = 0.28209479177387814
_SH_C1 = 0.4886025119029199
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435)


def _sh_basis_cpu(deg: int, d: torch.Tensor) -> torch.Tensor:
    """Real SH basis Y_k(d), d [..., 3] unit vectors -> [..., (deg+1)^2], in the 3DGS ordering and sign convention
    (gs-simp/utils/sh_utils.py:57-113). Plain PyTorch for CPU tensors (the gloo tests); GPU tensors go through
    the HIP kernel."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    b = [torch.full_like(x, _SH_C0)]
    if deg > 0:
        b += [-_SH_C1 * y, _SH_C1 * z, -_SH_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [_SH_C2[0] * xy, _SH_C2[1] * yz, _SH_C2[2] * (2 * zz - xx - yy), _SH_C2[3] * xz, _SH_C2[4] * (xx - yy)]
        if deg > 2:
            b += [_SH_C3[0] * y * (3 * xx - yy), _SH_C3[1] * xy * z, _SH_C3[2] * y * (4 * zz - xx - yy),
                  _SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), _SH_C3[4] * x * (4 * zz - xx - yy),
                  _SH_C3[5] * z * (xx - yy), _SH_C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=-1)


def sh_grad_from_factors(means3D, campos, factors, M: int, sh_degree: int, out=None):
    """dL/dSH summed over views from the views' colour factors: [P,3], [V,3], [V,P,3] -> [P,M,3]."""
    if means3D.is_cuda:
        from .raster import sh_backward_views
        return sh_backward_views(means3D, campos, factors, M, sh_degree, out=out)
    d = means3D[None] - campos[:, None]                                   # [V,P,3]
    d = d / d.norm(dim=-1, keepdim=True)
    Y = _sh_basis_cpu(sh_degree, d)                                       # [V,P,nb]
    res = torch.einsum("vpk,vpc->pkc", Y, factors)
    full = torch.zeros(means3D.shape[0], M, 3, dtype=means3D.dtype) if out is None else out.zero_()
    full[:, :res.shape[1]] = res
    return full


class FactoredGradExchange:
    """Per-step gradient exchange of view-parallel training with the SH gradient sent in factored form.

    Buffers (fp32, all torch-owned): `small` = [means3D 3 | opacities 1 | scales 3 | rotations 4] x P, all-reduced;
    `send` = [colour factor 3 x P | camera centre 3], all-gathered into `recv` [W, 3P + 3]; `shs` [P,M,3], rebuilt
    locally as the sum over the W views. `views` is what rasterize_backward(..., out=views, sh_grad="factor")
    writes into. After exchange() every rank holds the same summed gradients as GradBucket.all_reduce() would give
    (to fp32 summation order)."""

    SMALL = (("means3D", 3), ("opacities", 1), ("scales", 3), ("rotations", 4))

    def __init__(self, P: int, M: int, sh_degree: int, device, group=None):
        self.P, self.M, self.deg, self.group = P, M, sh_degree, group
        self.world = td.get_world_size(group)
        f32 = dict(dtype=torch.float32, device=device)
        self.small = torch.zeros(P * sum(w for _, w in self.SMALL), **f32)
        self.send = torch.zeros(3 * P + 3, **f32)
        self.recv = torch.zeros(self.world, 3 * P + 3, **f32)
        self.views: Dict[str, Optional[torch.Tensor]] = {}
        o = 0
        for n, w in self.SMALL:
            self.views[n] = self.small[o:o + P * w].view(P, w)
            o += P * w
        self.views["sh_color_factor"] = self.send[:3 * P].view(P, 3)
        self.views["means2D"] = torch.zeros(P, 3, **f32)
        self.shs = torch.zeros(P, M, 3, **f32)

    @staticmethod
    def pays(M: int, world: int) -> bool:
        """Factored beats one big all-reduce when (W-1)*3 < 2 (W-1)/W * 3M, i.e. W < 2M."""
        return M > 1 and world < 2 * M

    def begin_gather(self, campos: torch.Tensor):
        """Starts the all-gather of the colour factors (+ this rank's camera centre). Call as soon as
        views["sh_color_factor"] is final — with raster.rasterize_backward_split that is right after the render
        backward, so the transfer runs under the per-Gaussian chain rule."""
        self.send[3 * self.P:].copy_(campos.reshape(3).to(self.send.dtype))
        self._gather = td.all_gather_into_tensor(self.recv.view(-1), self.send, group=self.group, async_op=True)

    def finish(self, means3D: torch.Tensor):
        """All-reduces the other 11 floats per Gaussian, waits for the gather begun earlier and rebuilds dL/dSH."""
        td.all_reduce(self.small, op=td.ReduceOp.SUM, group=self.group)
        self._gather.wait()
        self._gather = None
        factors = self.recv[:, :3 * self.P].view(self.world, self.P, 3)
        sh_grad_from_factors(means3D, self.recv[:, 3 * self.P:], factors, self.M, self.deg, out=self.shs)
        g = {n: self.views[n] for n, _ in self.SMALL}
        g["shs"] = self.shs
        return g

    def exchange(self, means3D: torch.Tensor, campos: torch.Tensor):
        """means3D [P,3] (replicated parameters), campos [3] = this rank's camera centre. Returns the gradient dict
        {means3D, shs, opacities, scales, rotations} summed over ranks (views of the internal buffers)."""
        self.send[3 * self.P:].copy_(campos.reshape(3).to(self.send.dtype))
        h = td.all_gather_into_tensor(self.recv.view(-1), self.send, group=self.group, async_op=True)
        td.all_reduce(self.small, op=td.ReduceOp.SUM, group=self.group)
        h.wait()
        factors = self.recv[:, :3 * self.P].view(self.world, self.P, 3)
        sh_grad_from_factors(means3D, self.recv[:, 3 * self.P:], factors, self.M, self.deg, out=self.shs)
        g = {n: self.views[n] for n, _ in self.SMALL}
        g["shs"] = self.shs
        return g


class CompactedGradExchange(FactoredGradExchange):
    """FactoredGradExchange that moves only the rows (Gaussians) inside the GRADIENT SUPPORT of at least one rank. The render
    backward marks every Gaussian whose accumulation row it touches (mvi_raster_views.grad_support; RasterState.tensor(
    "grad_support")); a Gaussian outside that set — culled, outside the image, or simply occluded in that view — has an exactly
    zero gradient there, so neither its 11 floats nor its colour factor need the wire. In the bench scene the support of one
    1920 x 1080 view is 3 % of the 1.5 M Gaussians (87 % are "visible" in the radii > 0 sense, but the first ~240 entries of a
    tile's list saturate its pixels), so the union over 8 views is at most a quarter of the rows.

        1. the per-rank supports travel as BIT masks in one all-gather (P / 8 bytes per rank: 188 KB at P = 1.5 M; round 4:
           an all-reduce(MAX) of P bytes) and are OR-ed locally -> the union, identical on all ranks
        2. one scan of the union into a row list (csrc/compact_rows.hip); its length n stays on the device and is copied to
           pinned host memory on the side
        3. ONE gather launch packs the union rows of the four small gradients, the colour factors and the positions into
           buffers of `capacity` rows (capacity = 1.25 x the PREVIOUS step's n, known to the host without waiting), the
           all-reduce / all-gather run on those, dL/dSH is rebuilt for them, ONE scatter launch writes everything back
        4. the host then looks at n — an event far upstream of the work just queued, so the device never idles for it — and
           queues further pages of `capacity` rows if the union outgrew the buffers (exact in every case; round 3 read n back
           in the middle, built the buffers with torch.cat and scattered with five index_put launches)
       Unions above THRESHOLD of the rows take the full-size exchange of the base class.

    Wire per GPU: capacity / P x the base class's 241 MB (W = 8, M = 16, P = 1.5 M), plus the mask: ~35 MB at a union of 10 %.
    Costs: the mask's all-reduce (latency-bound), plan + gather + scatter + one zero-fill pass over the compacted rows. Passing a visibility filter
    (radii > 0) instead of the support is valid too (any superset of the support is), but compacts little: the union of 8
    views' visibility is 94 % of this scene. The sums are those of FactoredGradExchange (same elements; rows outside the
    union are exact zeros on both paths)."""

    THRESHOLD = 0.8
    HEADROOM = 1.25            # capacity of a step's buffers = HEADROOM x the previous step's union, rounded up to ROUND rows
    MIN_CAPACITY = 16384
    ROUND = 4096

    def __init__(self, P: int, M: int, sh_degree: int, device, group=None, split_sh: bool = False):
        """split_sh: the summed SH gradient is delivered as two tensors, "shs_dc" [P, 1, 3] and "shs_rest" [P, M - 1, 3] — the
        shapes of the reference's _features_dc / _features_rest parameters (gaussian_model.py:108-111), which the optimizer steps
        separately — instead of one "shs" [P, M, 3]; the scatter kernel writes both from the one compact array."""
        super().__init__(P, M, sh_degree, device, group=group)
        self.split_sh = bool(split_sh)
        if self.split_sh:
            f32 = dict(dtype=torch.float32, device=device)
            self.shs = None                                        # (allocated on demand by the full-size fallback)
            self.shs_dc, self.shs_rest = torch.zeros(P, 1, 3, **f32), torch.zeros(P, max(M - 1, 0), 3, **f32)
        self.last_union_fraction = None
        self.last_compacted = None
        self.last_pages = None
        self.last_capacity = None
        self._shs_rows = None          # rows of self.shs that may be non-zero: None = none, "all", an index tensor (CPU path) or
                                       # "plan" (GPU path: the row list of the previous step's plan, self._ws[self._cur ^ 1])
        self._cap = None               # GPU path: rows the compact buffers hold (None: not allocated yet)
        self._next_cap = P             # first step: nothing known about the union yet

    # ---- GPU path: nothing in the exchange waits for this step's row count -------------------------------------------------
    def _alloc(self, cap: int):
        dev, f32 = self.small.device, dict(dtype=torch.float32, device=self.small.device)
        self._cap = cap
        self._small_c = torch.zeros(11 * cap, **f32)               # SoA segments [means3D 3 | opacities 1 | scales 3 | rotations 4] x cap
        self._send_c = torch.zeros(3 * cap + 3, **f32)
        self._recv_c = torch.zeros(self.world * (3 * cap + 3), **f32)
        self._means_c = torch.zeros(cap, 3, **f32)
        self._sh_c = torch.zeros(cap, self.M, 3, **f32)
        if getattr(self, "_ws", None) is None:
            from . import _lib
            nbytes = _lib.lib().mvi_compact_workspace_bytes(self.P)
            self._ws_bytes = nbytes
            self._ws = [torch.empty(nbytes + 256, dtype=torch.uint8, device=dev) for _ in range(2)]   # plan + its count, ping-pong
            self._cur = 0
            self._mask = torch.zeros(self.P, dtype=torch.uint8, device=dev)
            self._count_pin = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._count_ev = torch.cuda.Event()

    def _window(self, scatter: bool, pairs, ws, first: int, cap: int):
        """One launch of csrc/compact_rows.hip's window kernel over (full [P, w], compact [cap, w]) tensor pairs."""
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        tab = (_lib.CompactTensor * len(pairs))()
        for e, pair in zip(tab, pairs):
            full, comp = pair[0], pair[1]
            off, stride = (pair[2], pair[3]) if len(pair) > 2 else (0, 0)      # words into a row of the compact side / between its rows
            src, dst = (comp, full) if scatter else (full, comp)
            addr = lambda t, is_comp: None if t is None else t.data_ptr() + (4 * off if is_comp else 0)
            e.in_, e.out, e.width, e.packed_stride = addr(src, scatter), addr(dst, not scatter), full[0].numel(), stride
        dev = self.small.device
        fn = L.mvi_compact_scatter_window if scatter else L.mvi_compact_gather_window
        with torch.cuda.device(dev):
            _lib.check(fn(tab, len(pairs), self.P, C.c_void_p(ws.data_ptr() + self._ws_bytes), first, cap, C.c_void_p(ws.data_ptr()),
                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "compact window")

    def _sh_pairs(self, sh_c, cap):
        """(full, compact[, offset, stride]) entries that scatter the compact dL/dSH rows [cap, M, 3] into the full-size output(s);
        sh_c = None: entries that zero those rows."""
        M3 = 3 * self.M
        if not self.split_sh:
            return [(self.shs.view(self.P, M3), None if sh_c is None else sh_c.view(cap, M3))]
        out = [(self.shs_dc.view(self.P, 3), None if sh_c is None else sh_c.view(cap, M3), 0, M3)]
        if self.M > 1:
            out.append((self.shs_rest.view(self.P, M3 - 3), None if sh_c is None else sh_c.view(cap, M3), 3, M3))
        return out

    def _result(self):
        g = {name: self.views[name] for name, _ in self.SMALL}
        if self.split_sh:
            g["shs_dc"], g["shs_rest"] = self.shs_dc, self.shs_rest
        else:
            g["shs"] = self.shs
        return g

    def _full_size(self, means3D, campos):
        """The base class's full-size exchange; split_sh: its [P, M, 3] result copied into the two outputs."""
        if self.split_sh and self.shs is None:
            self.shs = torch.zeros(self.P, self.M, 3, dtype=torch.float32, device=self.small.device)
        FactoredGradExchange.exchange(self, means3D, campos)
        if self.split_sh:
            self.shs_dc.copy_(self.shs[:, :1])
            self.shs_rest.copy_(self.shs[:, 1:])
        return self._result()

    def _union_mask(self, visible):
        """self._mask <- the union over ranks of the supports (GPU path). One all-gather of bit masks (MVI_DIST_BYTE_MASK=1: the
        round-4 form, an all-reduce(MAX) of the byte mask, for A/B runs)."""
        import ctypes as C
        import os
        from . import _lib
        P, dev = self.P, self.small.device
        flags = visible.to(torch.uint8).reshape(-1)
        if os.environ.get("MVI_DIST_BYTE_MASK") == "1":
            self._mask.copy_(flags)
            td.all_reduce(self._mask, op=td.ReduceOp.MAX, group=self.group)
            return
        if getattr(self, "_bits", None) is None:
            words = (P + 31) // 32
            self._bits = torch.zeros(words, dtype=torch.int32, device=dev)
            self._bits_all = torch.zeros(self.world * words, dtype=torch.int32, device=dev)
        flags = flags if flags.is_contiguous() else flags.contiguous()
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mvi_support_pack_bits(C.c_void_p(flags.data_ptr()), P, C.c_void_p(self._bits.data_ptr()), st), "support_pack_bits")
        td.all_gather_into_tensor(self._bits_all, self._bits, group=self.group)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mvi_support_union_bits(C.c_void_p(self._bits_all.data_ptr()), self.world, P,
                                                         C.c_void_p(self._mask.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                       "support_union_bits")

    def _page(self, means3D, campos, ws, first: int, cap: int):
        """Rows [first, first + cap) of the union list: gather, exchange, rebuild dL/dSH, scatter back. Buffers are cut to `cap`
        rows; rows past the list's end (the device knows where it ends) carry stale values that travel and are never scattered."""
        names = [name for name, _ in self.SMALL]
        widths = [w for _, w in self.SMALL]
        segs, o = [], 0
        for w in widths:
            segs.append(self._small_c[o:o + cap * w].view(cap, w))
            o += cap * w
        send = self._send_c[:3 * cap + 3]
        recv = self._recv_c[:self.world * (3 * cap + 3)].view(self.world, 3 * cap + 3)
        fac_c, means_c, sh_c = send[:3 * cap].view(cap, 3), self._means_c[:cap], self._sh_c[:cap]
        self._window(False, [(self.views[nm], sg) for nm, sg in zip(names, segs)] +
                     [(self.views["sh_color_factor"], fac_c), (means3D, means_c)], ws, first, cap)
        send[3 * cap:].copy_(campos.reshape(3).to(send.dtype))
        h = td.all_gather_into_tensor(recv.view(-1), send, group=self.group, async_op=True)
        td.all_reduce(self._small_c[:11 * cap], op=td.ReduceOp.SUM, group=self.group)
        h.wait()
        sh_grad_from_factors(means_c, recv[:, 3 * cap:], recv[:, :3 * cap].view(self.world, cap, 3), self.M, self.deg, out=sh_c)
        self._window(True, [(self.views[nm], sg) for nm, sg in zip(names, segs)] + self._sh_pairs(sh_c, cap), ws, first, cap)

    def _exchange_device(self, means3D: torch.Tensor, campos: torch.Tensor, visible: torch.Tensor):
        import ctypes as C
        from . import _lib
        P, dev = self.P, self.small.device
        cap = max(1, min(P, int(self._next_cap)))
        if self._cap is None or cap > self._cap or 2 * cap < self._cap:
            self._alloc(cap)
        self._cur ^= 1
        ws, ws_prev = self._ws[self._cur], self._ws[self._cur ^ 1]
        self._union_mask(visible)
        cnt = ws[self._ws_bytes:self._ws_bytes + 4].view(torch.int32)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mvi_compact_plan(C.c_void_p(self._mask.data_ptr()), P, C.c_void_p(ws.data_ptr()), self._ws_bytes,
                                                   C.c_void_p(cnt.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                       "compact_plan")
        self._count_pin.copy_(cnt, non_blocking=True)            # the host learns the count while the device works on page 0
        self._count_ev.record()
        # dL/dSH is zero outside the rows this step writes: clear what the previous step wrote (its row list is still in ws_prev)
        if self._shs_rows == "plan":
            self._window(True, self._sh_pairs(None, P), ws_prev, 0, P)
        elif self._shs_rows is not None:
            for t in ((self.shs_dc, self.shs_rest) if self.split_sh else (self.shs,)):
                t.zero_()
        self._page(means3D, campos, ws, 0, cap)
        self._count_ev.synchronize()                             # upstream of everything queued above: the device is not idle
        n = int(self._count_pin.item())
        pages = max(1, -(-n // cap))
        for p in range(1, pages):                                # the union outgrew the capacity chosen from the previous step
            self._page(means3D, campos, ws, p * cap, cap)
        self._shs_rows = "plan"
        self.last_union_fraction, self.last_compacted, self.last_pages, self.last_capacity = (n / P if P else 0.0), True, pages, cap
        self._next_cap = min(P, max(self.MIN_CAPACITY, -(-int(self.HEADROOM * n) // self.ROUND) * self.ROUND))
        return self._result()

    def exchange_visible(self, means3D: torch.Tensor, campos: torch.Tensor, visible: torch.Tensor):
        """visible [P] bool / uint8: any superset of this rank's gradient support — RasterState.tensor("grad_support", ...)
        after the backward (tight), or the visibility filter radii > 0 (gaussian_renderer/__init__.py:100; loose). Returns
        the same dict as exchange(). GPU tensors: the union rows travel in buffers sized from the PREVIOUS step's union
        (x HEADROOM), nothing waits for this step's count; if the union outgrew them, further pages follow (exact either
        way). Unions above THRESHOLD of the rows (previous step) take the full-size exchange of the base class."""
        P = self.P
        if visible.is_cuda:
            if self.last_union_fraction is not None and self.last_union_fraction > self.THRESHOLD:
                # full-size exchange; the union is still measured (on the side) so that a later step can compact again
                if getattr(self, "_mask", None) is None:
                    self._alloc(max(1, self.MIN_CAPACITY))
                self._union_mask(visible)
                frac = self._mask.float().mean()
                out = self._full_size(means3D, campos)
                self._shs_rows = "all"
                self.last_union_fraction, self.last_compacted, self.last_pages = float(frac.item()), False, 0
                self._next_cap = P
                return out
            return self._exchange_device(means3D, campos, visible)
        mask = visible.to(torch.uint8).contiguous().clone()
        td.all_reduce(mask, op=td.ReduceOp.MAX, group=self.group)
        names = [name for name, _ in self.SMALL]
        idx = mask.nonzero().squeeze(1)
        parts = [self.views[nm][idx] for nm in names]
        fac_c, means_c = self.views["sh_color_factor"][idx], means3D[idx]
        n = int(idx.numel())                                   # identical on every rank: derived from the reduced mask
        self.last_union_fraction = n / P if P else 0.0
        self.last_compacted = bool(P) and n <= self.THRESHOLD * P
        if not self.last_compacted:
            self._shs_rows = "all"
            return self._full_size(means3D, campos)
        widths = [w for _, w in self.SMALL]
        rows = torch.cat(parts, 1).contiguous()                                                       # [n, 11]
        send = torch.cat([fac_c.reshape(-1), campos.reshape(3).to(rows.dtype)])
        recv = torch.empty(self.world, 3 * n + 3, dtype=rows.dtype, device=rows.device)
        h = td.all_gather_into_tensor(recv.view(-1), send, group=self.group, async_op=True)
        td.all_reduce(rows, op=td.ReduceOp.SUM, group=self.group)
        h.wait()
        sh_c = sh_grad_from_factors(means_c.contiguous(), recv[:, 3 * n:].contiguous(),
                                    recv[:, :3 * n].reshape(self.world, n, 3).contiguous(), self.M, self.deg)
        # the full-size outputs are zero outside the rows written here: the small arrays are the backward's own outputs
        # (zero outside this rank's support, which the union contains), self.shs is cleared where the last step wrote it
        outs = (self.shs_dc, self.shs_rest) if self.split_sh else (self.shs,)
        for t in outs:
            if isinstance(self._shs_rows, str):
                t.zero_()
            elif self._shs_rows is not None:
                t[self._shs_rows] = 0
        if self.split_sh:
            self.shs_dc[idx] = sh_c[:, :1]
            self.shs_rest[idx] = sh_c[:, 1:]
        else:
            self.shs[idx] = sh_c
        self._shs_rows = idx
        g, o = {}, 0
        for (name, w) in self.SMALL:
            v = self.views[name]
            v[idx] = rows[:, o:o + w]
            g[name] = v
            o += w
        assert o == sum(widths)
        return self._result()

    exchange_support = exchange_visible


class RangedGradExchange(FactoredGradExchange):
    """FactoredGradExchange whose 11 small floats per Gaussian are exchanged in `n_ranges` pieces, each all-reduced as soon as
    the chain rule of its Gaussian range has been queued (raster.rasterize_backward_ranged drives it through
    mvi_raster_backward_geom_range): the collective of range r runs on RCCL's stream under the kernels of ranges r+1 ...,
    so only the last range's all-reduce (1 / n_ranges of the 44 P bytes) is exposed behind the backward, plus the SH rebuild.
    The buffer is range-major — range r holds [means3D 3 | opacities 1 | scales 3 | rotations 4] x n_r contiguously — so
    every piece is ONE contiguous all-reduce. Sums are identical to FactoredGradExchange (same elements, same reduction)."""

    def __init__(self, P: int, M: int, sh_degree: int, device, n_ranges: int = 4, group=None):
        super().__init__(P, M, sh_degree, device, group=group)
        per = -(-P // max(1, n_ranges))
        per = -(-per // 64) * 64                               # range starts are multiples of 64 (the kernels' block of Gaussians)
        self.ranges = [(a, min(per, P - a)) for a in range(0, P, per)] if P else []
        self.width = sum(w for _, w in self.SMALL)             # 11
        # Every sub-array of a range segment starts on a 16-byte boundary: the chain-rule kernel stores dL/drotations as
        # float4. Range starts are multiples of 64 rows, so only the LAST range can have a row count that is not a multiple
        # of 4; its sub-arrays are laid out for n rounded up to 4 rows (the pad rows stay zero and ride in the all-reduce).
        pad = (-P) % 4
        self.small = torch.zeros(self.width * (P + pad), dtype=torch.float32, device=device)
        self._range_views, self._range_span = [], []
        for first, n in self.ranges:
            npad = -(-n // 4) * 4
            seg, o, v = self.small[self.width * first:self.width * first + self.width * npad], 0, {}
            for name, w in self.SMALL:
                v[name] = seg[o:o + n * w].view(n, w)
                o += npad * w
            self._range_views.append(v)
            self._range_span.append((self.width * first, self.width * npad))
        # the SoA views of the base class do not describe this layout: assembled by finish()
        for name, _ in self.SMALL:
            self.views[name] = None
        self._works = []

    def range_views(self, r: int):
        """{means3D, opacities, scales, rotations} -> [n_r, w] tensors the chain rule of range r writes."""
        return self._range_views[r]

    def reduce_range(self, r: int):
        """Starts the all-reduce of range r (call right after its chain-rule kernel has been queued)."""
        o, length = self._range_span[r]
        self._works.append(td.all_reduce(self.small[o:o + length], op=td.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self, means3D: torch.Tensor):
        """Waits for the pieces and the gather begun earlier, rebuilds dL/dSH; returns {means3D, opacities, scales,
        rotations, shs} as [P, w] tensors (the small ones assembled from the range-major buffer: one 44 P-byte copy)."""
        for w in self._works:
            w.wait()
        self._works = []
        self._gather.wait()
        self._gather = None
        factors = self.recv[:, :3 * self.P].view(self.world, self.P, 3)
        sh_grad_from_factors(means3D, self.recv[:, 3 * self.P:], factors, self.M, self.deg, out=self.shs)
        g = {name: torch.cat([v[name] for v in self._range_views], 0) if self._range_views
             else self.small.new_zeros(0, w) for name, w in self.SMALL}
        g["shs"] = self.shs
        return g

    def exchange(self, means3D, campos):
        raise NotImplementedError("RangedGradExchange is driven range by range: begin_gather / reduce_range / finish")


# ---------------------------------------------------------------------------------------------------------------------
# Sharded Adam (reduce-scatter -> local step on the owned Gaussians -> all-gather of the updated rows)

class ShardPlan:
    """Rows (Gaussians) of every parameter tensor split over the ranks: rank r owns rows [r * rows, (r + 1) * rows) of the
    BODY = world * rows rows, `rows` a multiple of 64 (the kernels' block of Gaussians); the TAIL [body, P) — fewer than
    64 * world + 63 rows — is kept replicated: its gradient is all-reduced and every rank steps it redundantly. With that,
    any P shards with equal pieces and no padding, so the collectives work in place on the parameter tensors themselves."""

    def __init__(self, P: int, world: int, rank: int):
        self.P, self.world, self.rank = P, world, rank
        self.rows = (P // world) // 64 * 64
        self.body = self.rows * world
        self.tail = P - self.body
        self.first = rank * self.rows

    def own(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.first:self.first + self.rows]

    def tail_of(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.body:]


class ShardedAdam:
    """Adam over the reference's parameter groups (gaussian_model.py:154-163: one group per tensor, its own lr, eps 1e-15)
    for view-parallel training, with optimizer state and the step itself sharded by Gaussian (ShardPlan):

        step(grads):   grads[name] = THIS rank's view gradient, [P, ...]
          1. reduce-scatter of every gradient's body -> the summed gradient of the owned rows; all-reduce of the tail rows
          2. the inner optimizer (train_ops.FusedAdam on the GPU) steps the owned rows and the tail: 1 / world of the
             28 B / parameter of Adam traffic, 1 / world of the moments
          3. in-place all-gather of the owned rows of every parameter (input = the parameter's own slice)

    After step() every rank holds the same parameters as `inner` would give un-sharded on the summed gradients
    (bit-identical when the reduction order is: same elements, same kernel). step_factored() is the same with the
    SH gradient travelling as colour factors (FactoredGradExchange): owners receive the factor slices of all views by
    all-to-all and rebuild dL/dSH for their rows only.

    Wire per GPU and step, f = floats per Gaussian (11 + 3M), ring collectives: dense all-reduce + replicated step
    2 (W-1)/W * 4 f P; this class (W-1)/W * 4 f P (reduce-scatter) + (W-1)/W * 4 f P (all-gather) = the same bytes, minus
    (W-1)/W of the optimizer's HBM traffic. Against FactoredGradExchange it LOSES wire at high SH degree: the parameter
    all-gather moves all 3M SH floats, the factored gradient exchange only 3 (W-1) — 384 MB against 241 MB at W = 8,
    M = 16, P = 1.5 M — so bench.py keeps the replicated FusedAdam there and this is for sh_degree 0 / 1 models and for
    fitting more Gaussians (moments are 8 B / parameter). DESIGN.md §6 has the link arithmetic.

    Densification and checkpoints edit / store full-size moments (gaussian_model.py:335-404, :61-93): full_state()
    gathers them into torch.optim.Adam's layout, load_full_state() re-shards after the tensors were rebuilt."""

    def __init__(self, named_params: Dict[str, torch.Tensor], lrs, betas=(0.9, 0.999), eps=1e-15, group=None, inner=None):
        self.group = group
        self.world, self.rank = td.get_world_size(group), td.get_rank(group)
        self.betas, self.eps = betas, eps
        if inner is None:
            from .train_ops import FusedAdam as inner          # HIP kernel; CPU tests pass torch.optim.Adam
        self._inner_cls = inner
        self._lrs = {n: float(lrs[n] if isinstance(lrs, dict) else lrs) for n in named_params}
        self._bind(named_params)

    # -- construction ------------------------------------------------------------------------------------------
    def _bind(self, named_params):
        self.params = dict(named_params)
        Ps = {t.shape[0] for t in self.params.values()}
        if len(Ps) != 1:
            raise ValueError("ShardedAdam: every parameter tensor needs one row per Gaussian")
        for n, t in self.params.items():
            if not t.is_contiguous():
                raise ValueError(f"ShardedAdam: parameter {n} is not contiguous")
        self.plan = ShardPlan(Ps.pop(), self.world, self.rank)
        self._own = {n: self.plan.own(t.data) for n, t in self.params.items()}
        self._tail = {n: self.plan.tail_of(t.data) for n, t in self.params.items()}
        groups = []
        for n in self.params:
            ps = [t for t in (self._own[n], self._tail[n]) if t.numel()]
            groups.append({"params": ps, "lr": self._lrs[n], "name": n})
        self.inner = self._inner_cls([g for g in groups if g["params"]], lr=0.0, betas=self.betas, eps=self.eps)
        self._gown = {n: torch.zeros_like(self._own[n]) for n in self.params}

    def set_lr(self, name: str, lr: float):
        """The reference re-computes the xyz learning rate every iteration (gaussian_model.py:165-171)."""
        self._lrs[name] = float(lr)
        for g in self.inner.param_groups:
            if g["name"] == name:
                g["lr"] = float(lr)

    # -- the step ----------------------------------------------------------------------------------------------
    def _reduce_dense(self, name: str, grad: torch.Tensor, works):
        pl, w = self.plan, self.params[name][0].numel() if self.plan.P else 0
        g = grad if grad.is_contiguous() else grad.contiguous()
        if pl.rows:
            works.append(td.reduce_scatter_tensor(self._gown[name].view(-1), g.view(-1)[:pl.body * w], op=td.ReduceOp.SUM,
                                                  group=self.group, async_op=True))
        if pl.tail:
            tail = g[pl.body:].clone()
            works.append(td.all_reduce(tail, op=td.ReduceOp.SUM, group=self.group, async_op=True))
            return tail
        return None

    def _apply(self, tails):
        for n in self.params:
            if self._own[n].numel():
                self._own[n].grad = self._gown[n]
            if self._tail[n].numel():
                self._tail[n].grad = tails[n]
        self.inner.step()
        works = []
        pl = self.plan
        if pl.rows:
            for n, t in self.params.items():
                w = t[0].numel()
                full = t.data.view(-1)[:pl.body * w]
                src = self._own[n].reshape(-1)
                if td.get_backend(self.group) == "gloo":
                    src = src.clone()                      # gloo copies input -> output slot; keep it off the aliasing path
                works.append(td.all_gather_into_tensor(full, src, group=self.group, async_op=True))
        for h in works:
            h.wait()

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor]):
        works, tails = [], {}
        for n in self.params:
            tails[n] = self._reduce_dense(n, grads[n], works)
        for h in works:
            h.wait()
        self._apply(tails)

    @torch.no_grad()
    def step_factored(self, grads: Dict[str, torch.Tensor], sh_color_factor: torch.Tensor, campos: torch.Tensor,
                      means3D: torch.Tensor, sh_degree: int, dc_name: str = "f_dc", rest_name: str = "f_rest"):
        """grads: the non-SH gradients of this rank's view ([P, w] each, keys = the other parameter names);
        sh_color_factor [P, 3] (mvi_raster_backward(..., sh_grad="factor")), campos [3] this view's camera centre,
        means3D [P, 3] the (replicated) positions the view directions are taken from — call BEFORE they are stepped,
        i.e. pass the tensor the backward used. The SH gradient of the owned rows is rebuilt from all views' factors."""
        pl, W = self.plan, self.world
        works, tails = [], {}
        for n in self.params:
            if n not in (dc_name, rest_name):
                tails[n] = self._reduce_dense(n, grads[n], works)
        fac = sh_color_factor if sh_color_factor.is_contiguous() else sh_color_factor.contiguous()
        cams = torch.zeros(W, 3, dtype=fac.dtype, device=fac.device)
        works.append(td.all_gather_into_tensor(cams.view(-1), campos.reshape(3).to(fac.dtype).contiguous(), group=self.group,
                                               async_op=True))
        recv = torch.zeros(W, pl.rows, 3, dtype=fac.dtype, device=fac.device)
        if pl.rows:
            works.append(td.all_to_all_single(recv.view(-1), fac.view(-1)[:pl.body * 3], group=self.group, async_op=True))
        tail_fac = torch.zeros(W, pl.tail, 3, dtype=fac.dtype, device=fac.device)
        if pl.tail:
            works.append(td.all_gather_into_tensor(tail_fac.view(-1), fac[pl.body:].reshape(-1).clone(), group=self.group,
                                                   async_op=True))
        for h in works:
            h.wait()
        M = 1 + (self.params[rest_name].shape[1] if rest_name in self.params else 0)
        own_means = means3D[pl.first:pl.first + pl.rows]
        if pl.rows:
            sh = sh_grad_from_factors(own_means.contiguous(), cams, recv, M, sh_degree)
            self._gown[dc_name].copy_(sh[:, :1].reshape(self._gown[dc_name].shape))
            if rest_name in self.params:
                self._gown[rest_name].copy_(sh[:, 1:].reshape(self._gown[rest_name].shape))
        if pl.tail:
            sh_t = sh_grad_from_factors(means3D[pl.body:].contiguous(), cams, tail_fac, M, sh_degree)
            tails[dc_name] = sh_t[:, :1].reshape(self._tail[dc_name].shape).contiguous()
            if rest_name in self.params:
                tails[rest_name] = sh_t[:, 1:].reshape(self._tail[rest_name].shape).contiguous()
        else:
            tails[dc_name] = None
            tails[rest_name] = None
        self._apply(tails)

    # -- full-size state for densification / checkpoints ----------------------------------------------------------
    @torch.no_grad()
    def full_state(self) -> Dict[str, Dict[str, torch.Tensor]]:
        """{name: {"step", "exp_avg", "exp_avg_sq"}} with full [P, ...] moments on every rank (torch.optim.Adam's layout)."""
        pl, out = self.plan, {}
        for n, t in self.params.items():
            st_o = self.inner.state.get(self._own[n], {}) if self._own[n].numel() else {}
            st_t = self.inner.state.get(self._tail[n], {}) if self._tail[n].numel() else {}
            any_st = st_o or st_t
            entry = {"step": (any_st["step"].clone() if any_st else torch.tensor(0.0))}
            for key in ("exp_avg", "exp_avg_sq"):
                full = torch.zeros_like(t.data)
                if pl.rows:
                    src = st_o[key] if st_o else torch.zeros_like(self._own[n])
                    td.all_gather_into_tensor(full.view(-1)[:pl.body * t[0].numel()], src.reshape(-1).clone(), group=self.group)
                if pl.tail and st_t:
                    full[pl.body:] = st_t[key]
                entry[key] = full
            out[n] = entry
        return out

    @torch.no_grad()
    def load_full_state(self, named_params: Dict[str, torch.Tensor], state: Dict[str, Dict[str, torch.Tensor]]):
        """Re-binds to (possibly rebuilt, differently sized) parameter tensors and shards full-size moments onto the owners."""
        self._bind(named_params)
        pl = self.plan
        for n in self.params:
            if n not in state:
                continue
            for view, sl in ((self._own[n], slice(pl.first, pl.first + pl.rows)), (self._tail[n], slice(pl.body, pl.P))):
                if view.numel():
                    self.inner.state[view] = {"step": state[n]["step"].clone(),
                                              "exp_avg": state[n]["exp_avg"][sl].clone().contiguous(),
                                              "exp_avg_sq": state[n]["exp_avg_sq"][sl].clone().contiguous()}
