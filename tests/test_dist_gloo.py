"""world_size-2 gloo tests (CPU) of the view-parallel exchange step (multiview_inpaint_amd/dist.py).
The rasterizer itself needs the GPU; here the per-view gradients are stand-ins and the checks are on
the exchange: bucket sum == sum of per-rank gradients, identical on every rank, densification stats
reduced as per-view norms."""
import os
import socket

import torch
import torch.distributed as td
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    from multiview_inpaint_amd import dist as md
    P, M = 1000, 4
    views = list(range(7))
    mine = md.shard_views(views, rank, world)
    assert mine == [v for v in views if v % world == rank]
    # 1. GradBucket: one flat all-reduce equals the sum of the per-rank SoA gradients
    bucket = md.GradBucket(P, M, "cpu")
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    per_rank = [{k: torch.randn(v.shape, generator=gens[r]) for k, v in sorted(bucket.views.items()) if k != "means2D"}
                for r in range(world)]
    for k, v in per_rank[rank].items():
        bucket.views[k].copy_(v)
    bucket.all_reduce()
    for k in per_rank[0]:
        want = sum(per_rank[r][k] for r in range(world))
        assert torch.allclose(bucket.views[k], want, rtol=1e-5, atol=1e-6), k
    assert bucket.flat.numel() == P * (11 + 3 * M)
    # 2. parameter-gradient exchange over the reference's six parameter groups
    shapes = [(P, 3), (P, 1, 3), (P, M - 1, 3), (P, 1), (P, 3), (P, 4)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    g2 = [torch.Generator().manual_seed(200 + r) for r in range(world)]
    grads = [[torch.randn(s, generator=g2[r]) for s in shapes] for r in range(world)]
    for p, g in zip(params, grads[rank]):
        p.grad = g.clone()
    if rank == 1:
        params[3].grad = None                             # a rank whose view saw nothing for that group
    md.all_reduce_param_grads(params)
    for i, p in enumerate(params):
        want = sum(grads[r][i] for r in range(world) if not (r == 1 and i == 3))
        assert torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6), i
    # 3. densification side channel: per-view norms are summed, radii are max-ed
    g3 = [torch.Generator().manual_seed(300 + r) for r in range(world)]
    vg = [torch.randn(P, 3, generator=g3[r]) for r in range(world)]
    vis = [torch.rand(P, generator=g3[r]) > 0.4 for r in range(world)]
    rad = [torch.randint(0, 50, (P,), generator=g3[r], dtype=torch.int32) for r in range(world)]
    accum, denom, mx = torch.zeros(P, 1), torch.zeros(P, 1), torch.zeros(P)
    md.reduce_densification_stats(vg[rank], vis[rank], rad[rank], accum, denom, mx)
    want_a = sum(torch.where(vis[r][:, None], vg[r][:, :2].norm(dim=-1, keepdim=True), torch.zeros(P, 1)) for r in range(world))
    want_d = sum(vis[r].float()[:, None] for r in range(world))
    want_m = torch.stack([torch.where(vis[r], rad[r].float(), torch.zeros(P)) for r in range(world)]).max(0).values
    assert torch.allclose(accum, want_a, atol=1e-6) and torch.equal(denom, want_d) and torch.equal(mx, want_m)
    # ... and with reduce="mean" (the backward ran on loss / world): norm_scale = world restores the per-view norm of the
    # UNSCALED gradient, the statistic densify_grad_threshold is compared with (gaussian_model.py:467-470, :482-484)
    accum2, denom2, mx2 = torch.zeros(P, 1), torch.zeros(P, 1), torch.zeros(P)
    md.reduce_densification_stats(vg[rank] / world, vis[rank], rad[rank], accum2, denom2, mx2, norm_scale=float(world))
    assert torch.allclose(accum2, want_a, atol=1e-5) and torch.equal(denom2, want_d) and torch.equal(mx2, want_m)
    # 4. factored SH exchange == all-reduce of the dense gradients (SH part rebuilt from 3 floats per view)
    for Mx, deg in ((16, 3), (16, 1), (4, 1)):
        g4 = [torch.Generator().manual_seed(400 + 10 * Mx + r) for r in range(world)]
        means = torch.randn(P, 3, generator=torch.Generator().manual_seed(7)) * 2 + torch.tensor([0.0, 0.0, 5.0])
        cams = [torch.randn(3, generator=g4[r]) for r in range(world)]
        fac = [torch.randn(P, 3, generator=g4[r]) * (torch.rand(P, 1, generator=g4[r]) > 0.3) for r in range(world)]
        small = [{n: torch.randn(P, w, generator=g4[r]) for n, w in md.FactoredGradExchange.SMALL} for r in range(world)]
        ex = md.FactoredGradExchange(P, Mx, deg, "cpu")
        ex.views["sh_color_factor"].copy_(fac[rank])
        for n, v in small[rank].items():
            ex.views[n].copy_(v)
        got = ex.exchange(means, cams[rank])
        nb = (deg + 1) ** 2
        dense = torch.zeros(P, Mx, 3)
        for r in range(world):
            d = means - cams[r]
            Y = md._sh_basis_cpu(deg, d / d.norm(dim=-1, keepdim=True))           # [P, nb]
            dense[:, :nb] += Y[:, :, None] * fac[r][:, None, :]
        assert torch.allclose(got["shs"], dense, rtol=1e-5, atol=1e-6), (Mx, deg)
        assert float(got["shs"][:, nb:].abs().max() if nb < Mx else 0.0) == 0.0
        for n, _ in md.FactoredGradExchange.SMALL:
            assert torch.allclose(got[n], sum(small[r][n] for r in range(world)), rtol=1e-5, atol=1e-6), n
        # the overlapped form (gather begun before the other gradients exist) gives the same sums
        ex2 = md.FactoredGradExchange(P, Mx, deg, "cpu")
        ex2.views["sh_color_factor"].copy_(fac[rank])
        ex2.begin_gather(cams[rank])
        for n, v in small[rank].items():                  # "the chain rule" finishes after the gather has started
            ex2.views[n].copy_(v)
        got2 = ex2.finish(means)
        assert torch.equal(got2["shs"], got["shs"])
        for n, _ in md.FactoredGradExchange.SMALL:
            assert torch.equal(got2[n], got[n]), n
    assert md.FactoredGradExchange.pays(16, 8) and not md.FactoredGradExchange.pays(4, 8) and not md.FactoredGradExchange.pays(1, 2)
    torch.save(dict(flat=bucket.flat, accum=accum, shs=got["shs"]), os.path.join(out_dir, f"r{rank}.pt"))
    td.destroy_process_group()


def test_view_parallel_exchange_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = (torch.load(tmp_path / f"r{r}.pt") for r in range(world))
    assert torch.equal(a["flat"], b["flat"]) and torch.equal(a["accum"], b["accum"])   # ranks agree bit for bit
    assert torch.equal(a["shs"], b["shs"])


def test_sh_basis_matches_reference_eval_sh_golden():
    """dist._sh_basis_cpu against the imported reference's eval_sh (tests/golden/raster_partial.npz, generated by
    tools/gen_golden_raster.py from gs-simp/utils/sh_utils.py:57-112): sum_k Y_k(d) sh[k] + 0.5, clamped at 0."""
    import numpy as np
    from multiview_inpaint_amd import dist as md
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raster_partial.npz"))
    dirs, sh = torch.tensor(G["sh_dirs"]), torch.tensor(G["sh_coeffs"])
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    for deg in range(4):
        Y = md._sh_basis_cpu(deg, dirs)                                           # [64, nb]
        rgb = torch.clamp_min(torch.einsum("pk,pkc->pc", Y, sh[:, :Y.shape[1]]) + 0.5, 0.0)
        assert torch.allclose(rgb, torch.tensor(G[f"sh_rgb_deg{deg}"]), rtol=1e-5, atol=1e-6), deg


def _oracle_view_gradients(rank, world):
    """Per-rank gradients of a DIFFERENT camera view of ONE small scene, from the CPU oracle (the checker, allowed in
    tests): what each rank's rasterizer backward hands to the exchange in view-parallel training."""
    import numpy as np
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from multiview_inpaint_amd import synthetic as syn
    from oracle import raster_oracle as ro
    from raster_helpers import oracle_params
    rng = np.random.default_rng(5)
    Rm, T0 = syn.random_rotation(rng), rng.normal(size=3)
    cams = [syn.make_camera(96, 64, 50.0, Rm, T0 + np.array([0.3 * v, -0.15 * v, 0.05 * v])) for v in range(world)]
    sc = syn.make_scene(500, cams[0], 3, 5, log_scale_mean=np.log(0.06), zmin=1.0, zmax=6.0)
    bg = np.array([0.2, 0.3, 0.1], np.float32)
    out = []
    for v, cam in enumerate(cams):
        p = oracle_params(ro, cam, sc, bg)
        kw = dict(shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
        f = ro.forward(p, sc["means3D"], sc["opacities"], **kw)
        g_img = np.random.default_rng(50 + v).normal(size=(3, cam["H"], cam["W"])).astype(np.float32)
        b = ro.backward(p, f, g_img, sc["means3D"], **kw)
        out.append((cam, b, f))
    return sc, out[rank], out


def _worker_oracle(rank, world, port, out_dir):
    """The rank-1 identity of the SH gradient, END TO END on real per-view gradients: each rank puts the oracle's gradients
    of ITS view into the factored exchange (colour factor = dL/dSH[:, 0, :] / C0, the 11 small floats as they are); what
    comes out must be the sum over ranks of the DENSE per-view gradients, dL/dSH included."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from multiview_inpaint_amd import dist as md
    sc, (cam, b, f), every = _oracle_view_gradients(rank, world)
    P, M, deg = sc["means3D"].shape[0], sc["shs"].shape[1], 3
    assert int((f["radii"] > 0).sum()) > P // 3                      # this view sees a good part of the scene
    C0 = 0.28209479177387814
    ex = md.FactoredGradExchange(P, M, deg, "cpu")
    ex.views["sh_color_factor"].copy_(torch.tensor(b["shs"][:, 0, :] / C0))
    for n, _ in md.FactoredGradExchange.SMALL:
        ex.views[n].copy_(torch.tensor(b[n]).reshape(ex.views[n].shape))
    got = ex.exchange(torch.tensor(sc["means3D"]), torch.tensor(cam["campos"]))
    for n in ("means3D", "opacities", "scales", "rotations", "shs"):
        want = sum(torch.tensor(e[1][n]).reshape(got[n].shape) for e in every)
        scale = float(want.abs().max())
        err = float((got[n] - want).abs().max())
        assert err <= 1e-5 * scale, (n, err, scale)                 # all-reduced == sum of the single-view gradients, 1e-5 rel
    # and the dense bucket gives the same sums
    bucket = md.GradBucket(P, M, "cpu")
    for n in ("means3D", "opacities", "scales", "rotations", "shs"):
        bucket.views[n].copy_(torch.tensor(b[n]).reshape(bucket.views[n].shape))
    bucket.all_reduce()
    assert float((bucket.views["shs"] - got["shs"]).abs().max()) <= 1e-5 * float(got["shs"].abs().max())
    torch.save(dict(shs=got["shs"].clone(), means3D=got["means3D"].clone()), os.path.join(out_dir, f"o{rank}.pt"))
    td.destroy_process_group()


def test_factored_exchange_of_real_per_view_gradients_world2(tmp_path):
    world = 2
    mp.spawn(_worker_oracle, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = (torch.load(tmp_path / f"o{r}.pt") for r in range(world))
    assert torch.equal(a["shs"], b["shs"]) and torch.equal(a["means3D"], b["means3D"])   # ranks agree bit for bit


def _worker_ranged(rank, world, port, out_dir):
    """RangedGradExchange (range-major buffer, one all-reduce per Gaussian range) gives the sums of FactoredGradExchange."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    from multiview_inpaint_amd import dist as md
    P, M, deg = 1000, 16, 3                                  # 1000 rows in 4 ranges: 256, 256, 256, 232 (64-aligned starts)
    g = [torch.Generator().manual_seed(900 + r) for r in range(world)]
    means = torch.randn(P, 3, generator=torch.Generator().manual_seed(3)) + torch.tensor([0.0, 0.0, 4.0])
    cams = [torch.randn(3, generator=g[r]) for r in range(world)]
    fac = [torch.randn(P, 3, generator=g[r]) for r in range(world)]
    small = [{n: torch.randn(P, w, generator=g[r]) for n, w in md.FactoredGradExchange.SMALL} for r in range(world)]
    ex = md.RangedGradExchange(P, M, deg, "cpu", n_ranges=4)
    assert [a for a, _ in ex.ranges] == [0, 256, 512, 768] and sum(n for _, n in ex.ranges) == P
    ex.views["sh_color_factor"].copy_(fac[rank])
    ex.begin_gather(cams[rank])
    for r, (first, n) in enumerate(ex.ranges):
        for name, v in ex.range_views(r).items():
            v.copy_(small[rank][name][first:first + n])
        ex.reduce_range(r)
    got = ex.finish(means)
    ref = md.FactoredGradExchange(P, M, deg, "cpu")
    ref.views["sh_color_factor"].copy_(fac[rank])
    for name, v in small[rank].items():
        ref.views[name].copy_(v)
    want = ref.exchange(means, cams[rank])
    for name in ("means3D", "opacities", "scales", "rotations", "shs"):
        assert got[name].shape == want[name].shape and torch.equal(got[name], want[name]), name
    td.destroy_process_group()


def test_ranged_exchange_equals_factored_exchange_world2(tmp_path):
    mp.spawn(_worker_ranged, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def _worker_sharded_adam(rank, world, port, out_dir):
    """ShardedAdam (reduce-scatter -> owners step -> in-place all-gather) against torch.optim.Adam on the summed gradients:
    ragged P (body 256 rows in 2 shards + 77 replicated tail rows), per-group learning rates that change between steps,
    the factored SH form, and the full-size state round trip densification needs (prune to another P, keep stepping)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    from multiview_inpaint_amd import dist as md
    P, M, deg = 333, 4, 1
    shapes = {"xyz": (P, 3), "f_dc": (P, 1, 3), "f_rest": (P, M - 1, 3), "opacity": (P, 1), "scaling": (P, 3), "rotation": (P, 4)}
    lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}
    g0 = torch.Generator().manual_seed(5)
    init = {n: torch.randn(s, generator=g0) for n, s in shapes.items()}
    init["xyz"] = init["xyz"] + torch.tensor([0.0, 0.0, 5.0])
    plan = md.ShardPlan(P, world, rank)
    assert (plan.rows, plan.body, plan.tail) == (128, 256, 77)

    def make_ref(tensors):
        ps = {n: torch.nn.Parameter(t.clone()) for n, t in tensors.items()}
        return ps, torch.optim.Adam([{"params": [p], "lr": lrs[n], "name": n} for n, p in ps.items()], lr=0.0, eps=1e-15)

    ref_p, ref = make_ref(init)
    mine = {n: t.clone() for n, t in init.items()}
    opt = md.ShardedAdam(mine, lrs, eps=1e-15, inner=torch.optim.Adam)
    gens = [torch.Generator().manual_seed(50 + r) for r in range(world)]
    for it in range(3):
        per_rank = [{n: torch.randn(s, generator=gens[r]) for n, s in shapes.items()} for r in range(world)]
        if it == 1:
            opt.set_lr("xyz", 9e-5)
            [g for g in ref.param_groups if g["name"] == "xyz"][0]["lr"] = 9e-5
        opt.step(per_rank[rank])
        for n, p in ref_p.items():
            p.grad = sum(per_rank[r][n] for r in range(world))
        ref.step()
        for n in shapes:
            assert torch.equal(mine[n], ref_p[n].data), (it, n)
    # full-size state, as torch.optim.Adam lays it out
    full = opt.full_state()
    for n, p in ref_p.items():
        assert float(full[n]["step"]) == 3.0
        assert torch.equal(full[n]["exp_avg"], ref.state[p]["exp_avg"]) and torch.equal(full[n]["exp_avg_sq"], ref.state[p]["exp_avg_sq"]), n
    # prune (the reference's _prune_optimizer on full tensors), re-shard, keep stepping: P 333 -> 201 (body 128, tail 73)
    keep = torch.rand(P, generator=torch.Generator().manual_seed(9)) > 0.4
    keep[torch.nonzero(keep)[201:]] = False
    P2 = int(keep.sum())
    assert P2 == 201
    pruned = {n: mine[n][keep].contiguous() for n in shapes}
    pstate = {n: {"step": full[n]["step"], "exp_avg": full[n]["exp_avg"][keep], "exp_avg_sq": full[n]["exp_avg_sq"][keep]} for n in shapes}
    opt.load_full_state(pruned, pstate)
    assert (opt.plan.rows, opt.plan.body, opt.plan.tail) == (64, 128, 73)
    ref_p2, ref2 = make_ref(pruned)
    [g for g in ref2.param_groups if g["name"] == "xyz"][0]["lr"] = 9e-5         # the sharded optimizer keeps its learning rates
    for n, p in ref_p2.items():
        ref2.state[p] = {"step": torch.tensor(3.0), "exp_avg": pstate[n]["exp_avg"].clone(), "exp_avg_sq": pstate[n]["exp_avg_sq"].clone()}
    per_rank = [{n: torch.randn((P2,) + s[1:], generator=gens[r]) for n, s in shapes.items()} for r in range(world)]
    opt.step(per_rank[rank])
    for n, p in ref_p2.items():
        p.grad = sum(per_rank[r][n] for r in range(world))
    ref2.step()
    for n in shapes:
        assert torch.equal(pruned[n], ref_p2[n].data), n
    # factored SH form == dense form on the rank-1 SH gradients it stands for
    a = {n: t.clone() for n, t in init.items()}
    b = {n: t.clone() for n, t in init.items()}
    oa = md.ShardedAdam(a, lrs, eps=1e-15, inner=torch.optim.Adam)
    ob = md.ShardedAdam(b, lrs, eps=1e-15, inner=torch.optim.Adam)
    cams = [torch.randn(3, generator=gens[r]) for r in range(world)]
    fac = [torch.randn(P, 3, generator=gens[r]) * (torch.rand(P, 1, generator=gens[r]) > 0.3) for r in range(world)]
    small = [{n: torch.randn(shapes[n], generator=gens[r]) for n in ("xyz", "opacity", "scaling", "rotation")} for r in range(world)]
    dense_sh = md.sh_grad_from_factors(init["xyz"], cams[rank][None], fac[rank][None], M, deg)      # this view's dense dL/dSH
    dense = dict(small[rank], f_dc=dense_sh[:, :1].contiguous(), f_rest=dense_sh[:, 1:].contiguous())
    oa.step(dense)
    ob.step_factored(small[rank], fac[rank], cams[rank], init["xyz"], deg)
    for n in shapes:
        assert torch.allclose(a[n], b[n], rtol=1e-5, atol=1e-7), n
    torch.save({n: t for n, t in b.items()}, os.path.join(out_dir, f"s{rank}.pt"))
    td.destroy_process_group()


def test_sharded_adam_world2(tmp_path):
    mp.spawn(_worker_sharded_adam, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(tmp_path / f"s{r}.pt") for r in range(2))
    for n in a:
        assert torch.equal(a[n], b[n]), n                       # replicas stay bit-identical


def _worker_compacted(rank, world, port, out_dir):
    """World 4, ragged P (not a multiple of anything), rank 2's view sees NOTHING: the visibility-compacted exchange against
    the plain sums, on the compacted path (views that overlap little) and on the full-size path it falls back to."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    from multiview_inpaint_amd import dist as md
    P, M, deg = 1237, 16, 3
    means = torch.randn(P, 3, generator=torch.Generator().manual_seed(7)) * 2 + torch.tensor([0.0, 0.0, 5.0])
    results = {}
    for case, p_vis in (("sparse", 0.18), ("dense", 0.7)):
        g = [torch.Generator().manual_seed(900 + 10 * r + (0 if case == "sparse" else 5)) for r in range(world)]
        vis = [(torch.rand(P, generator=g[r]) < p_vis) if r != 2 else torch.zeros(P, dtype=torch.bool) for r in range(world)]
        cams = [torch.randn(3, generator=g[r]) for r in range(world)]
        fac = [torch.randn(P, 3, generator=g[r]) * vis[r][:, None] for r in range(world)]
        small = [{n: torch.randn(P, w, generator=g[r]) * vis[r][:, None] for n, w in md.FactoredGradExchange.SMALL}
                 for r in range(world)]
        ex = md.CompactedGradExchange(P, M, deg, "cpu")
        ex.views["sh_color_factor"].copy_(fac[rank])
        for n, v in small[rank].items():
            ex.views[n].copy_(v)
        got = ex.exchange_visible(means, cams[rank], vis[rank])
        union = torch.stack(vis).any(0)
        assert ex.last_union_fraction == int(union.sum()) / P
        assert ex.last_compacted == (case == "sparse"), (case, ex.last_union_fraction)
        dense = torch.zeros(P, M, 3)
        for r in range(world):
            d = means - cams[r]
            Y = md._sh_basis_cpu(deg, d / d.norm(dim=-1, keepdim=True))
            dense += Y[:, :, None] * fac[r][:, None, :]
        assert torch.allclose(got["shs"], dense, rtol=1e-5, atol=1e-6), case
        assert float(got["shs"][~union].abs().max()) == 0.0
        for n, _ in md.FactoredGradExchange.SMALL:
            assert torch.allclose(got[n], sum(small[r][n] for r in range(world)), rtol=1e-5, atol=1e-6), (case, n)
            assert float(got[n][~union].abs().max()) == 0.0
        results[case] = {k: v.clone() for k, v in got.items()}
    # view sharding with more views than ranks and an uneven remainder
    assert md.shard_views(list(range(10)), rank, world) == [v for v in range(10) if v % world == rank]
    torch.save(results, os.path.join(out_dir, f"c{rank}.pt"))
    td.destroy_process_group()


def test_visibility_compacted_exchange_world4_ragged_with_an_empty_view(tmp_path):
    world = 4
    mp.spawn(_worker_compacted, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"c{r}.pt") for r in range(world)]
    for case in ("sparse", "dense"):
        for k in res[0][case]:
            assert all(torch.equal(res[0][case][k], res[r][case][k]) for r in range(1, world)), (case, k)   # ranks agree bit for bit
