"""View-parallel multi-GPU reconstruction: one process per GPU, Gaussian parameters
replicated, every rank rasterizes a different camera view, ONE all-reduce (RCCL over xGMI; gloo in
the CPU tests) of a flat gradient bucket per step, plus the small side-channel reductions that keep
densification (gs-simp/scene/gaussian_model.py:466-484, gs-simp/train.py:113-116) identical on
every rank. The reference itself has no distributed 3DGS code (process per scene,
gs-simp/train.sh:1); its single-GPU loop is gs-simp/inpaint_rec.py:71-172.

No collective runs inside the rasterizer: the exchange step is the gradient sum only.
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
                               group=None):
    """Multi-view counterpart of add_densification_stats (gaussian_model.py:482-484) + the max-radii
    update (train.py:115): each rank contributes the norm of ITS view's screen-space gradient where
    ITS view saw the Gaussian; sums / max are taken over ranks so every rank holds identical stats."""
    norm = torch.zeros_like(xyz_gradient_accum)
    norm[visibility] = torch.norm(viewspace_grad[visibility, :2], dim=-1, keepdim=True)
    cnt = visibility.to(denom.dtype).view_as(denom).clone()
    rad = torch.where(visibility, radii.to(max_radii2D.dtype), torch.zeros_like(max_radii2D))
    pack = torch.cat([norm.reshape(-1), cnt.reshape(-1)])
    td.all_reduce(pack, op=td.ReduceOp.SUM, group=group)
    td.all_reduce(rad, op=td.ReduceOp.MAX, group=group)
    n = norm.numel()
    xyz_gradient_accum += pack[:n].view_as(xyz_gradient_accum)
    denom += pack[n:].view_as(denom)
    torch.maximum(max_radii2D, rad, out=max_radii2D)
