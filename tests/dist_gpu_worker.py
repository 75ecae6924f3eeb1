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


# ---- the view-sharded TRAINING loop (BASELINE configs[4] at two ranks) through the SHIPPED module, multiview_inpaint_amd.train_views:
# ViewShardedTrainer on a stand-in of the reference's GaussianModel (tests/gs_standin.py) with dropin.patch_gs_simp's surgery hooks
# installed on it — stored parameters into the raw-parameter rasterizer, fused L1 + DSSIM incl. the mask of the non-inpainted views,
# backward into the exchange's buffers, bit-mask all-gather + compacted exchange, reduced densification statistics, densify_and_prune
# on every rank with an identically seeded generator, FusedAdam — against ONE process that renders every view of a step with the same
# pieces (world = 1), sums the gradients and runs the same tail.
def run_training_rank(rank, world, port, N=20_000, W=320, H=208, deg=2, steps=14, say=print):
    import numpy as np
    import torch
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from multiview_inpaint_amd import raster as R, synthetic as syn, train_ops as T, train_views as TV
        from multiview_inpaint_amd.dropin import patch_gs_simp
        import gs_standin as GS
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        sc = syn.make_scene(N, syn.make_camera(W, H, 50.0), deg, seed=0, log_scale_mean=np.log(0.04))

        class Model(GS.StandinGaussianModel):                        # the hooks an unchanged script gets (patch_gs_simp.install)
            prune_points = patch_gs_simp._make_prune_points(GS.StandinGaussianModel.prune_points)
            cat_tensors_to_optimizer = patch_gs_simp._make_cat_tensors(GS.StandinGaussianModel.cat_tensors_to_optimizer)

        def target_image(k):                                         # what view k should look like: the scene with other colours
            g = torch.Generator(dev).manual_seed(77)
            shs = torch.tensor(sc["shs"], device=dev)
            shs[:, 0] += 0.6 * torch.randn(N, 3, device=dev, generator=g)
            t0 = {k2: torch.tensor(sc[k2], device=dev) for k2 in ("means3D", "opacities", "scales", "rotations")}
            cam = camera(syn, np, k, W, H)
            rs = R.GaussianRasterizationSettings(
                image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
                scale_modifier=1.0, viewmatrix=torch.tensor(cam["viewmatrix"], device=dev), projmatrix=torch.tensor(cam["projmatrix"], device=dev),
                sh_degree=deg, campos=torch.tensor(cam["campos"], device=dev), prefiltered=False)
            img, _, _, _ = R.rasterize_forward(rs, t0["means3D"], t0["opacities"], shs=shs, scales=t0["scales"], rotations=t0["rotations"])
            return img

        cams = []
        for k in range(6):                                           # six views; every third one is a NON-inpainted view with a mask
            mask = None
            if k % 3 == 2:
                mask = torch.zeros(1, H, W, device=dev)
                mask[:, H // 4:H // 2, W // 3:2 * W // 3] = 1.0
            cams.append(GS.StandinCamera(camera(syn, np, k, W, H), target_image(k), mask=mask, inpainted=mask is None, name=f"v{k}"))
        opt_args = GS.StandinOpt(iterations=steps + 1, densify_from_iter=3, densify_until_iter=steps + 1, densification_interval=5,
                                 densify_grad_threshold=2e-5)      # one densification inside the run (iteration 5 and 10)
        bg = torch.zeros(3, device=dev)
        extent = 6.0

        # (1) this rank's share of the sharded run
        m = Model(sc, deg, optimizer_cls=T.FusedAdam)
        tr = TV.ViewShardedTrainer(m, opt_args, lambda: list(cams), bg, extent, seed=3)
        assert tr.world == world and tr.rank == rank
        shard_losses, shard_P = [], []
        stat_sum = None
        for it in range(1, steps + 1):
            loss3 = tr.step(it)
            tot = loss3[:1].detach().clone().cpu()
            td.all_reduce(tot)                                       # the step's loss summed over the views (for the comparison only)
            shard_losses.append(float(tot))
            shard_P.append(int(m._xyz.shape[0]))
            if it == 4:                                              # the statistic the first densification (iteration 5) will threshold
                stat_sum = (m.xyz_gradient_accum / m.denom.clamp_min(1)).clone()
        same = tr.replicas_identical()
        # (1b) the same run with reduce="mean" (ADVICE r5): the optimizer steps on the MEAN of the view gradients (Adam: the same
        # step up to eps), but the densification statistic must stay the per-view norm of the UNSCALED gradient — otherwise
        # densify_grad_threshold is silently multiplied by world and densification all but stops
        mm = Model(sc, deg, optimizer_cls=T.FusedAdam)
        trm = TV.ViewShardedTrainer(mm, opt_args, lambda: list(cams), bg, extent, seed=3, reduce="mean")
        mean_P, stat_mean = [], None
        for it in range(1, steps + 1):
            trm.step(it)
            mean_P.append(int(mm._xyz.shape[0]))
            if it == 4:
                stat_mean = (mm.xyz_gradient_accum / mm.denom.clamp_min(1)).clone()
        stat_err = float((stat_mean - stat_sum).abs().max() / stat_sum.abs().max())
        mean_ok = stat_err < 1e-3 and all(abs(a - b) <= max(2, 0.01 * b) for a, b in zip(mean_P, shard_P)) and trm.replicas_identical()
        # (2) one process, every view of a step, summed gradients, the same tail
        m1 = Model(sc, deg, optimizer_cls=T.FusedAdam)
        t1 = TV.ViewShardedTrainer(m1, opt_args, lambda: list(cams), bg, extent, seed=3, world=1)
        stack = TV.ViewStack(lambda: list(cams), world, 0, seed=3)     # the draws of the sharded run
        names = [n for n, _ in TV.PARAMS]
        key = dict(xyz="xyz", opacity="opacity", scaling="scaling", rotation="rotation", f_dc="features_dc", f_rest="features_rest")
        single_losses, single_P = [], []
        for it in range(1, steps + 1):
            m1.update_learning_rate(it)
            p = t1.params()
            acc, tot = None, 0.0
            for cam in stack.next_views():
                st, rs, radii, loss3, g = t1.view_gradients(cam, it)
                acc = {n: g[key[n]].clone() for n in names} if acc is None else {n: acc[n] + g[key[n]] for n in names}
                tot += float(loss3[0])
                if it < opt_args.densify_until_iter:
                    t1._densification_stats(g["means2D"], radii)
            t1.finish_step(it, p, acc)
            single_losses.append(tot)
            single_P.append(int(m1._xyz.shape[0]))
        torch.cuda.synchronize()
        ok = same and mean_ok
        first_change = next((i for i in range(steps) if single_P[i] != N), steps)
        # before the first change of P the two runs see the same rows: loss curves to fp32 summation order; after it a Gaussian
        # whose accumulated statistic sits on the densification threshold may be split in one run and not in the other
        worst_pre = max(abs(a - b) / abs(b) for a, b in zip(shard_losses[:first_change + 1], single_losses[:first_change + 1]))
        worst_post = max([abs(a - b) / abs(b) for a, b in zip(shard_losses[first_change + 1:], single_losses[first_change + 1:])] or [0.0])
        ok &= worst_pre < 1e-4 and worst_post < 2e-2
        ok &= first_change < steps and shard_P[-1] != N                       # the densification happened, in both runs
        ok &= all(abs(a - b) <= max(2, 0.01 * b) for a, b in zip(shard_P, single_P))
        # (no "the loss goes down" check: cloning 12 k Gaussians with their full opacity raises the loss of the next steps — it is the
        # single-process run doing exactly the same that is asserted)
        every = [None] * world
        td.all_gather_object(every, shard_P)
        ok &= all(e == every[0] for e in every)                               # every rank went through the same sizes
        say(f"rank {rank}: loss (sum over {world} views) {shard_losses[0]:.5f} -> {shard_losses[-1]:.5f}; P {N} -> {shard_P[-1]} (single process "
            f"{single_P[-1]}), first change at iteration {first_change + 1}; sharded vs single-process loss curve {worst_pre:.2e} before it, "
            f"{worst_post:.2e} after; replicas identical after {steps} steps incl. densify + prune: {same}; reduce='mean': densification "
            f"statistic before the first densification equals reduce='sum' to {stat_err:.1e}, P {N} -> {mean_P[-1]}: {mean_ok}")
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


# ---- the LAUNCHER itself (multiview_inpaint_amd.train_views.main) under torchrun with two ranks over gloo, on the stand-in gs-simp
# directory tests/gs_simp_standin: argument plumbing, device pinning, patching, rank-0 render sets / report / saves, the loop, the
# replicas check and the exit code. Runs in a process forked by the clean forkserver (it starts a program: torchrun).
def entry_launcher(rank, world, port, queue, kwargs):
    import glob
    import subprocess
    import tempfile
    lines = []
    try:
        out = tempfile.mkdtemp(prefix="mvi_launcher_")
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), os.environ.get("PYTHONPATH", "")]),
                   MVI_TRAIN_VIEWS_BACKEND="gloo", MVI_TRAIN_VIEWS_DEVICES="0,0")    # both ranks pin themselves to the one GPU
        for var in ("MVI_TRAIN_VIEWS_NO_PIN", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            env.pop(var, None)
        n = int(kwargs.get("iterations", 12))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "-m", "multiview_inpaint_amd.train_views", os.path.join(ROOT, "tests", "gs_simp_standin"),
               "-m", out, "--scene_id", "standin", "--n_mode", "2", "--dry-run", str(n), "--reduce", kwargs.get("reduce", "sum"),
               "--checkpoint_iterations", str(n)]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
        lines += ["$ " + " ".join(cmd[1:]), p.stdout[-3000:], p.stderr[-3000:]]
        base = os.path.join(out, "2")
        found = dict(
            renders0=len(glob.glob(os.path.join(base, "ours_0", "renders", "*.png"))), gt0=len(glob.glob(os.path.join(base, "ours_0", "gt", "*.png"))),
            rendersN=len(glob.glob(os.path.join(base, f"ours_{n}", "renders", "*.png"))),
            saved=os.path.exists(os.path.join(base, f"point_cloud/iteration_{n}", "point_cloud.pt")),
            ckpt=os.path.exists(os.path.join(base, f"chkpnt{n}.pth")))
        lines.append(f"launcher rc {p.returncode}; files: {found}")
        ok = (p.returncode == 0 and found == dict(renders0=4, gt0=4, rendersN=6, saved=True, ckpt=True)
              and "Evaluating test: L1" in p.stdout and "Evaluating train: L1" in p.stdout and "Training complete." in p.stdout
              and "patched:" in p.stderr and "the replicas diverged" not in p.stderr)
        queue.put((rank, ok, [], lines))
    except BaseException as e:
        import traceback
        queue.put((rank, False, [], lines + [traceback.format_exc(), repr(e)]))
