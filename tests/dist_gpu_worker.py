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


# ---- the view-sharded TRAINING step (BASELINE configs[4] at two ranks): render own view -> L1 + DSSIM gradient -> backward into the
# exchange's views -> compacted exchange -> Adam on the replicated parameters, against ONE process that renders every view of a step,
# sums the gradients and takes the same Adam step.
def run_training_rank(rank, world, port, N=20_000, W=320, H=208, deg=2, steps=6, say=print):
    import numpy as np
    import torch
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multiview_inpaint_amd import dist as md, raster as R, synthetic as syn, train_ops as T
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        M = (deg + 1) ** 2
        sc = syn.make_scene(N, syn.make_camera(W, H, 50.0), deg, seed=0, log_scale_mean=np.log(0.04))
        names = ("means3D", "opacities", "scales", "rotations", "shs")
        lrs = dict(means3D=1.6e-4, opacities=1e-2, scales=1e-3, rotations=1e-3, shs=2.5e-3)

        def settings(k):
            cam = camera(syn, np, k, W, H)
            return R.GaussianRasterizationSettings(
                image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
                scale_modifier=1.0, viewmatrix=torch.tensor(cam["viewmatrix"], device=dev), projmatrix=torch.tensor(cam["projmatrix"], device=dev),
                sh_degree=deg, campos=torch.tensor(cam["campos"], device=dev), prefiltered=False)

        def fresh():
            t = {k: torch.tensor(sc[k], device=dev) for k in names}
            opt = T.FusedAdam([{"params": [t[k]], "lr": lrs[k], "name": k} for k in names], lr=0.0, eps=1e-15)
            return t, opt

        def target(k):                                    # what the views should look like: the scene with other colours
            g = torch.Generator(dev).manual_seed(77)
            shs = torch.tensor(sc["shs"], device=dev)
            shs[:, 0] += 0.6 * torch.randn(N, 3, device=dev, generator=g)
            t0 = {k2: torch.tensor(sc[k2], device=dev) for k2 in names}
            img, _, _, _ = R.rasterize_forward(settings(k), t0["means3D"], t0["opacities"], shs=shs, scales=t0["scales"], rotations=t0["rotations"])
            return img

        def view_gradient(t, k, out=None, sh_grad="dense"):
            rs = settings(k)
            kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
            img, _, _, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], prepare_backward=True, **kw)
            out3, g_img = T.photometric_loss_forward_backward(img, target(k), 0.2)
            g = R.rasterize_backward(rs, st, g_img, t["means3D"], out=out, sh_grad=sh_grad, **kw)
            return rs, st, g, out3[0]

        # (1) this rank's share of the sharded run
        t, opt = fresh()
        ex = md.CompactedGradExchange(N, M, deg, dev)
        ex.THRESHOLD, ex.MIN_CAPACITY, ex.ROUND = 1.0, 256, 256
        shard_losses = []
        for step in range(steps):
            rs, st, _, loss = view_gradient(t, (step % 2) * world + rank, out=ex.views, sh_grad="factor")   # 2 x world views, revisited
            got = ex.exchange_support(t["means3D"], rs.campos, st.tensor("grad_support", (N,), torch.uint8))
            for k in names:
                t[k].grad = got[k].view_as(t[k])
            opt.step()
            total = loss.detach().clone().cpu()
            td.all_reduce(total)                          # the step's loss summed over the views (for the comparison only)
            shard_losses.append(float(total))
        # (2) one process, every view of a step, summed gradients
        t1, opt1 = fresh()
        single_losses = []
        for step in range(steps):
            acc, tot = None, 0.0
            for k in range(world):
                _, _, g, loss = view_gradient(t1, (step % 2) * world + k)
                acc = {n: g[n].clone() for n in names} if acc is None else {n: acc[n] + g[n] for n in names}
                tot += float(loss)
            for n in names:
                t1[n].grad = acc[n].view_as(t1[n])
            opt1.step()
            single_losses.append(tot)
        torch.cuda.synchronize()
        ok = True
        worst = max(abs(a - b) / abs(b) for a, b in zip(shard_losses, single_losses))
        ok &= worst < 1e-4 and single_losses[-2] < single_losses[0] and single_losses[-1] < single_losses[1]     # (same views two visits later)
        rms = {}
        for n in names:
            d, ref = (t[n] - t1[n]).double(), (t1[n] - torch.tensor(sc[n], device=dev)).double()     # against the distance travelled
            rms[n] = float(d.pow(2).mean().sqrt() / (ref.pow(2).mean().sqrt() + 1e-30))
            ok &= rms[n] < 2e-2
        digest = torch.stack([t[n].double().sum() for n in names]).cpu()
        every = [torch.zeros_like(digest) for _ in range(world)]
        td.all_gather(every, digest)
        same = all(torch.equal(every[0], e) for e in every)           # the replicas stay bit-identical
        ok &= same
        say(f"rank {rank}: loss (sum over {world} views) {single_losses[0]:.5f} -> {single_losses[-2]:.5f} and {single_losses[1]:.5f} -> {single_losses[-1]:.5f}; sharded vs single-process loss curve "
            f"{worst:.2e}; parameter difference / distance travelled (rms) {max(rms.values()):.2e}; replicas identical: {same}")
        flag = torch.tensor([1.0 if ok else 0.0])
        td.all_reduce(flag, op=td.ReduceOp.MIN)
        return flag.item() == 1.0, []
    finally:
        td.destroy_process_group()


def entry_training(rank, world, port, queue, kwargs):
    lines = []
    try:
        ok, _ = run_training_rank(rank, world, port, say=lines.append, **kwargs)
        queue.put((rank, ok, [], lines))
    except BaseException as e:
        import traceback
        queue.put((rank, False, [], lines + [traceback.format_exc(), repr(e)]))
