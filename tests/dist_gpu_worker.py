"""W ranks of the view-sharded gradient exchange on ONE GPU over gloo (RCCL refuses two ranks on one GPU): the DEVICE path of
dist.CompactedGradExchange with more than one rank. Every rank renders its own view of one replicated scene, runs the backward into
the exchange's views and exchanges; every rank ALSO renders all W views itself and sums the gradients: the exchanged result must equal
that sum (to fp32 summation order), be identical on all ranks, and be exactly zero outside the union of the supports. Steps on one
exchange object: no history (capacity P), history-sized capacity with other views, a forced tiny capacity (paging), a step after it.
Shared by tests/test_dist_gpu_ranks.py (2 ranks, inside the GPU suite) and tools/experiments/dist_gpu_ranks.py (any W, by hand).
The processes that run `entry` must come from a parent that never initialised the GPU (a forkserver / spawn made beforehand)."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def camera(syn, np, k, W, H):
    a = math.radians(5.0 * k)
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    centre = np.array([0.0, 0.0, 4.5])
    pos = centre - R @ np.array([0.0, 0.0, 4.5])
    return syn.make_camera(W, H, 50.0, R, -R.T @ pos)


def run_rank(rank, world, port, N=200_003, W=640, H=400, deg=3, forced=(None, None, 3000, None), say=print):
    import numpy as np
    import torch
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multiview_inpaint_amd import dist as md, raster as R, synthetic as syn
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        M = (deg + 1) ** 2
        sc = syn.make_scene(N, syn.make_camera(W, H, 50.0), deg, seed=0)
        t = {k: torch.tensor(v, device=dev) for k, v in sc.items() if k != "sh_degree"}
        kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        ex = md.CompactedGradExchange(N, M, deg, dev)
        ex.THRESHOLD, ex.MIN_CAPACITY, ex.ROUND = 1.0, 256, 256
        ok, pages = True, []
        for step, force in enumerate(forced):
            def view(k):
                cam = camera(syn, np, k, W, H)
                rs = R.GaussianRasterizationSettings(
                    image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
                    scale_modifier=1.0, viewmatrix=torch.tensor(cam["viewmatrix"], device=dev),
                    projmatrix=torch.tensor(cam["projmatrix"], device=dev), sh_degree=deg, campos=torch.tensor(cam["campos"], device=dev),
                    prefiltered=False)
                g_img = torch.randn(3, H, W, device=dev, generator=torch.Generator(dev).manual_seed(1000 * step + k))
                return rs, g_img
            want, union = None, torch.zeros(N, dtype=torch.bool, device=dev)
            for k in range(world):                                    # the plain sum of every rank's view, computed locally
                rs, g_img = view(step * world + k)
                _, _, _, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], **kw)
                g = R.rasterize_backward(rs, st, g_img, t["means3D"], **kw)
                union |= st.tensor("grad_support", (N,), torch.uint8).bool()
                want = {n: g[n].clone() for n in ("means3D", "opacities", "scales", "rotations", "shs")} if want is None else \
                       {n: want[n] + g[n] for n in want}
            rs, g_img = view(step * world + rank)
            _, _, _, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], prepare_backward=True, **kw)
            R.rasterize_backward(rs, st, g_img, t["means3D"], out=ex.views, sh_grad="factor", **kw)
            if force is not None:
                ex._next_cap = force
            got = ex.exchange_support(t["means3D"], rs.campos, st.tensor("grad_support", (N,), torch.uint8))
            torch.cuda.synchronize()
            n = int(union.sum())
            errs = {}
            for name in want:
                a, b = got[name].double(), want[name].double()
                errs[name] = float((a - b).abs().max() / (b.abs().max() + 1e-30))
                ok &= errs[name] < 2e-5 and float(got[name][~union].abs().max()) == 0.0
            ok &= abs(ex.last_union_fraction * N - n) < 0.5
            digest = torch.stack([got[name].double().sum() for name in sorted(want)]).cpu()
            every = [torch.zeros_like(digest) for _ in range(world)]
            td.all_gather(every, digest)
            ok &= all(torch.equal(every[0], e) for e in every)        # every rank holds the same sums, bit for bit
            pages.append(ex.last_pages)
            say(f"step {step} rank {rank}: union {n} rows ({ex.last_union_fraction:.4f}), capacity {ex.last_capacity}, pages {ex.last_pages}, "
                f"max rel err vs the plain sum {max(errs.values()):.2e}, zero outside the union, ranks agree: {ok}")
        flag = torch.tensor([1.0 if ok else 0.0])
        td.all_reduce(flag, op=td.ReduceOp.MIN)
        return flag.item() == 1.0, pages
    finally:
        td.destroy_process_group()


def entry(rank, world, port, queue, kwargs):
    """Process target: reports (rank, ok, pages per step, lines) — or the exception text — through `queue`."""
    lines = []
    try:
        ok, pages = run_rank(rank, world, port, say=lines.append, **kwargs)
        queue.put((rank, ok, pages, lines))
    except BaseException as e:       # the parent must hear about it
        import traceback
        queue.put((rank, False, [], lines + [traceback.format_exc(), repr(e)]))
