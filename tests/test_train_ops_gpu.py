"""GPU parity of the training-loop ops (include/mvi_train_ops.h) through the C-ABI:
  * against the golden vectors of the imported reference (tests/golden/loss_small.npz) and the numpy oracle
    (oracle/loss_oracle.py) at small sizes — tolerance 1e-4 relative (north_star), in practice ~1e-6;
  * at 1920x1080 against a plain PyTorch fp32 restatement of the same formula run on the GPU, and through
    size-independent properties (identical images, scaling of the upstream gradient, mask semantics)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
G = np.load(os.path.join(ROOT, "tests", "golden", "loss_small.npz"))
CASES = sorted({k.split("_")[0] for k in G.files})


@pytest.fixture(scope="module")
def T():
    assert torch.cuda.is_available()
    from multiview_inpaint_amd import train_ops
    return train_ops


def _torch_loss(img, gt, lam, weight=None):
    """loss_utils.py:17-62 + train.py:91-92 restated with torch ops (differentiable, any device)."""
    import loss_oracle as lo
    w1 = torch.tensor(lo.window_1d(), device=img.device)
    win = (w1[:, None] @ w1[None, :]).expand(3, 1, 11, 11).contiguous()
    x, y = (img, gt) if weight is None else (img * weight, gt * weight)
    conv = lambda t: F.conv2d(t[None], win, padding=5, groups=3)[0]
    mu1, mu2 = conv(x), conv(y)
    s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    smap = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    return (1 - lam) * (x - y).abs().mean() + lam * (1 - smap.mean())


@pytest.mark.parametrize("name", CASES)
def test_loss_and_gradient_match_reference_golden(T, name):
    import loss_oracle as lo
    img, gt = torch.tensor(G[f"{name}_image"]).cuda(), torch.tensor(G[f"{name}_gt"]).cuda()
    mask = torch.tensor(G[f"{name}_mask"]).cuda() if f"{name}_mask" in G.files else None
    lam = float(G[f"{name}_lambda"])
    out3, grad = T.photometric_loss_forward_backward(img, gt, lam, None if mask is None else 1.0 - mask)
    out3 = out3.cpu().numpy()
    assert abs(out3[0] - float(G[f"{name}_loss"])) < 1e-5
    assert abs(out3[1] - float(G[f"{name}_l1"])) < 1e-5 and abs(out3[2] - float(G[f"{name}_ssim"])) < 1e-5
    g_ref = G[f"{name}_grad"].astype(np.float64)
    assert np.abs(grad.cpu().numpy() - g_ref).max() < 1e-4 * np.abs(g_ref).max()
    o = lo.photometric_loss(G[f"{name}_image"], G[f"{name}_gt"], lam, None if mask is None else 1.0 - G[f"{name}_mask"][0])
    assert np.abs(grad.cpu().numpy() - o["grad"]).max() < 2e-5 * np.abs(o["grad"]).max()
    assert abs(out3[0] - o["loss"]) < 2e-6


def test_reference_named_functions_are_differentiable(T):
    """l1_loss / ssim with the reference's names and the fused form give the same loss and the same image gradient."""
    name = "c"
    gt, mask = torch.tensor(G[f"{name}_gt"]).cuda(), torch.tensor(G[f"{name}_mask"]).cuda()
    img = torch.tensor(G[f"{name}_image"]).cuda().requires_grad_(True)
    pd, tg = img * (1.0 - mask), gt * (1.0 - mask)                       # inpaint_rec.py:120-123
    loss = 0.8 * T.l1_loss(pd, tg) + 0.2 * (1.0 - T.ssim(pd, tg))
    loss.backward()
    g_sep = img.grad.clone()
    img.grad = None
    fused = T.fused_l1_dssim_loss(img, gt, 0.2, mask=mask)
    (3.0 * fused).backward()
    assert abs(loss.item() - float(G[f"{name}_loss"])) < 1e-5 and abs(fused.item() - loss.item()) < 1e-6
    g_ref = torch.tensor(G[f"{name}_grad"]).cuda()
    assert float((g_sep - g_ref).abs().max()) < 1e-4 * float(g_ref.abs().max())
    assert float((img.grad / 3.0 - g_ref).abs().max()) < 1e-4 * float(g_ref.abs().max())


def test_l1_loss_then_ssim_on_the_same_tensors_share_one_node(T):
    """The reference's loss expression (train.py:91-92) through the reference-named functions: ssim(a, b) right after l1_loss(a, b)
    on the same tensor objects is the other output of ONE node (one statistics pass, one gradient pass weighted by the two upstream
    gradients on the device). Values identical to the stand-alone calls; gradient equal to the fused loss's; anything that breaks
    the match (other tensors, an in-place edit in between, a second ssim call) takes the stand-alone path with the same values."""
    name = "c"
    gt = torch.tensor(G[f"{name}_gt"]).cuda()
    img = torch.tensor(G[f"{name}_image"]).cuda().requires_grad_(True)
    calls = []
    orig = T._LossPair.apply
    T._LossPair.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        l1 = T.l1_loss(img, gt)
        ss = T.ssim(img, gt)
        assert len(calls) == 1 and ss.grad_fn is l1.grad_fn                       # ONE node
        loss = 0.8 * l1 + 0.2 * (1.0 - ss)
        loss.backward()
        g_pair, img.grad = img.grad.clone(), None
        fused = T.fused_l1_dssim_loss(img, gt, 0.2)
        fused.backward()
        g_fused, img.grad = img.grad.clone(), None
        assert abs(loss.item() - fused.item()) < 1e-6
        assert float((g_pair - g_fused).abs().max()) < 2e-6 * float(g_fused.abs().max())
        # the L1 value dropped before ssim is called (one expression): still one node
        n = len(calls)
        loss2 = 0.8 * T.l1_loss(img, gt) + 0.2 * (1.0 - T.ssim(img, gt))
        assert len(calls) == n + 1 and loss2.item() == loss.item() and T._pending_pair is None
        # stand-alone values (a second ssim call finds nothing pending)
        ss2 = T.ssim(img, gt)
        assert ss2.grad_fn is not l1.grad_fn and ss2.item() == ss.item()
        # L1 alone: the SSIM half gets no gradient
        l1b = T.l1_loss(img, gt)
        assert l1b.item() == l1.item()
        l1b.backward()
        want = torch.sign(img.detach() - gt) / img.numel()
        assert float((img.grad - want).abs().max()) < 1e-9
        img.grad = None
        # other tensor objects with the same values: no match, same numbers
        T.l1_loss(img, gt)
        ss3 = T.ssim(img * 1.0, gt)
        assert ss3.grad_fn is not l1.grad_fn and abs(ss3.item() - ss.item()) < 1e-7
        # an in-place edit between the two calls: no match (the pending values belong to the old contents)
        buf = img.detach().clone()
        l1c = T.l1_loss(buf, gt)
        buf.mul_(0.5)
        ss4 = T.ssim(buf, gt)
        assert abs(ss4.item() - ss.item()) > 1e-3 and l1c.item() == l1.item()
        # under no_grad (training_report, render.py): values only, nothing kept
        with torch.no_grad():
            a, b = T.l1_loss(img, gt), T.ssim(img, gt)
        assert a.item() == l1.item() and b.item() == ss.item() and not a.requires_grad
        # l1_loss under no_grad, ssim differentiated: the detached half must not be handed over (the SSIM term would lose its gradient)
        with torch.no_grad():
            T.l1_loss(img, gt)
        ss5 = T.ssim(img, gt)
        assert ss5.requires_grad and ss5.item() == ss.item()
        (1.0 - ss5).backward()
        assert float(img.grad.abs().max()) > 0
        img.grad = None
        # the L1 value backpropagated (graph freed) before ssim is called: ssim gets a node of its own instead of the freed one
        l1d = T.l1_loss(img, gt)
        l1d.backward()
        assert T._pending_pair is None
        img.grad = None
        ss6 = T.ssim(img, gt)
        ss6.backward()
        assert ss6.item() == ss.item() and float(img.grad.abs().max()) > 0
        img.grad = None
    finally:
        T._LossPair.apply = orig


def test_full_size_against_torch_restatement_and_properties(T):
    H, W = 1080, 1920
    g = torch.Generator(device="cuda").manual_seed(0)
    gt = torch.rand(3, H, W, device="cuda", generator=g)
    img = (gt + 0.1 * torch.randn(3, H, W, device="cuda", generator=g)).clamp(0, 1)
    mask = (torch.rand(1, H, W, device="cuda", generator=g) > 0.7).float()
    for wt in (None, 1.0 - mask):
        x = img.clone().requires_grad_(True)
        ref = _torch_loss(x, gt, 0.2, wt)
        ref.backward()
        out3, grad = T.photometric_loss_forward_backward(img, gt, 0.2, wt)
        assert abs(out3[0].item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
        assert float((grad - x.grad).abs().max()) < 1e-4 * float(x.grad.abs().max())
        if wt is not None:
            assert float(grad[:, mask[0] > 0].abs().max()) == 0.0       # masked pixels receive no gradient
    # identical images: l1 = 0, ssim = 1, loss = 0
    out3, grad = T.photometric_loss_forward_backward(gt, gt, 0.2)
    assert float(out3[1]) == 0.0 and abs(float(out3[2]) - 1.0) < 1e-6 and abs(float(out3[0])) < 1e-6
    # upstream scale is linear; the loss value is bit-reproducible
    o1, g1 = T.photometric_loss_forward_backward(img, gt, 0.2, upstream=1.0)
    o2, g2 = T.photometric_loss_forward_backward(img, gt, 0.2, upstream=2.0)
    assert torch.equal(o1, o2) and float((g2 - 2.0 * g1).abs().max()) < 1e-12 + 1e-6 * float(g1.abs().max())


def test_ragged_sizes_and_errors(T):
    for H, W in ((1, 1), (5, 3), (17, 31), (16, 33)):
        g = torch.Generator().manual_seed(H * 100 + W)
        gt, img = torch.rand(3, H, W, generator=g).cuda(), torch.rand(3, H, W, generator=g).cuda()
        x = img.clone().requires_grad_(True)
        ref = _torch_loss(x, gt, 0.2)
        ref.backward()
        out3, grad = T.photometric_loss_forward_backward(img, gt, 0.2)
        assert abs(out3[0].item() - ref.item()) < 1e-5
        assert float((grad - x.grad).abs().max()) < 1e-4 * float(x.grad.abs().max())
    with pytest.raises(ValueError):
        T.photometric_loss_forward_backward(torch.zeros(3, 4, 4).cuda(), torch.zeros(3, 4, 5).cuda())
    with pytest.raises(RuntimeError):
        T.photometric_loss_forward_backward(torch.zeros(3, 4, 4), torch.zeros(3, 4, 4))
    with pytest.raises(NotImplementedError):
        T.ssim(torch.zeros(3, 4, 4).cuda(), torch.zeros(3, 4, 4).cuda(), window_size=7)


def test_fused_adam_matches_torch_adam(T):
    """Six parameter groups as in gaussian_model.py:154-163 (per-group lr, eps 1e-15), 5 steps with changing lr:
    parameters and both moments equal torch.optim.Adam's (the reference's optimizer), state layout included."""
    torch.manual_seed(0)
    P = 4099
    shapes = dict(xyz=(P, 3), f_dc=(P, 1, 3), f_rest=(P, 15, 3), opacity=(P, 1), scaling=(P, 3), rotation=(P, 4))
    lrs = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=2.5e-3 / 20, opacity=0.05, scaling=5e-3, rotation=1e-3)
    init = {k: torch.randn(s) for k, s in shapes.items()}
    def make(opt_cls):
        ps = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in init.items()}
        return ps, opt_cls([{"params": [ps[k]], "lr": lrs[k], "name": k} for k in shapes], lr=0.0, eps=1e-15)
    pa, oa = make(torch.optim.Adam)
    pb, ob = make(T.FusedAdam)
    for it in range(5):
        g = torch.Generator().manual_seed(100 + it)
        grads = {k: (torch.randn(s, generator=g) * (0.0 if (k == "opacity" and it == 2) else 1e-2)).cuda() for k, s in shapes.items()}
        for ps, opt in ((pa, oa), (pb, ob)):
            for grp in opt.param_groups:
                if grp["name"] == "xyz":
                    grp["lr"] = lrs["xyz"] * (0.9 ** it)              # update_learning_rate, gaussian_model.py:169-175
            for k in shapes:
                ps[k].grad = grads[k].clone()
            opt.step()
            opt.zero_grad(set_to_none=True)
    for k in shapes:
        sa, sb = oa.state[pa[k]], ob.state[pb[k]]
        assert set(sb.keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sb["step"]) == 5.0
        for a, b, what in ((pa[k], pb[k], "param"), (sa["exp_avg"], sb["exp_avg"], "exp_avg"), (sa["exp_avg_sq"], sb["exp_avg_sq"], "exp_avg_sq")):
            err = float((a - b).abs().max() / (a.abs().max() + 1e-30))
            assert err < 2e-6, (k, what, err)
    # the state dict round-trips into torch.optim.Adam (checkpoint compatibility)
    pc, oc = make(torch.optim.Adam)
    oc.load_state_dict(ob.state_dict())
    assert float(oc.state[pc["xyz"]]["step"]) == 5.0
    with pytest.raises(RuntimeError):
        cpu_p = torch.nn.Parameter(torch.zeros(4))
        cpu_p.grad = torch.zeros(4)
        T.FusedAdam([cpu_p]).step()


def test_adam_ragged_sizes_and_many_tensors(T):
    torch.manual_seed(1)
    sizes = [1, 3, 5, 1023, 1024, 1025, 4097, 7, 2, 100003]           # 10 tensors -> two launches; unaligned tails
    pa = [torch.nn.Parameter(torch.randn(n).cuda()) for n in sizes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa, ob = torch.optim.Adam(pa, lr=1e-2), T.FusedAdam(pb, lr=1e-2)
    for it in range(3):
        for a, b in zip(pa, pb):
            a.grad = torch.randn_like(a)
            b.grad = a.grad.clone()
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        assert float((a - b).abs().max()) < 2e-6 * float(a.abs().max() + 1.0)


def test_activate_gaussians_forward_backward(T):
    torch.manual_seed(2)
    P, M = 3001, 16
    raw = dict(s=torch.randn(P, 3) - 4.0, r=torch.randn(P, 4), o=torch.randn(P, 1) * 2, dc=torch.randn(P, 1, 3), rest=torch.randn(P, M - 1, 3) * 0.2)
    raw["r"][5] = 0.0                                                     # degenerate quaternion: F.normalize's eps clamp
    def run(fn):
        t = {k: v.clone().cuda().requires_grad_(True) for k, v in raw.items()}
        outs = fn(t["s"], t["r"], t["o"], t["dc"], t["rest"])
        g = torch.Generator(device="cuda").manual_seed(3)
        cot = [torch.randn(o.shape, device="cuda", generator=g) for o in outs]
        torch.autograd.backward(outs, cot)
        return outs, {k: v.grad for k, v in t.items()}
    ref = lambda s, r, o, dc, rest: (torch.exp(s), F.normalize(r), torch.sigmoid(o), torch.cat((dc, rest), dim=1))   # gaussian_model.py:44-59, :95-115
    o_ref, g_ref = run(ref)
    o_hip, g_hip = run(T.activate_gaussians)
    for a, b in zip(o_ref, o_hip):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())
    for k in raw:
        a, b = g_ref[k], g_hip[k]
        ok = torch.ones(P, dtype=torch.bool, device="cuda")
        if k == "r":
            ok[5] = False                                              # 0/eps: both finite, direction arbitrary
        assert float((a[ok] - b[ok]).abs().max()) <= 1e-5 * float(a[ok].abs().max()), k
    assert torch.equal(o_hip[3][:, :1], raw["dc"].cuda()) and torch.equal(o_hip[3][:, 1:], raw["rest"].cuda())
    # degree 0: no features_rest
    s, r, o, shs = T.activate_gaussians(raw["s"].cuda(), raw["r"].cuda(), raw["o"].cuda(), raw["dc"].cuda(), torch.zeros(P, 0, 3).cuda())
    assert shs.shape == (P, 1, 3) and torch.equal(shs, raw["dc"].cuda())


@pytest.mark.parametrize("N", [4, 255, 256, 257, 5000])
def test_distcuda2_matches_exact_3nn(T, N):
    import loss_oracle as lo
    rng = np.random.default_rng(N)
    pts = (rng.normal(size=(N, 3)) * np.array([3.0, 1.0, 0.3])).astype(np.float32)
    if N >= 256:
        pts[10] = pts[200]                                             # a duplicated point: distance 0 counts
    got = T.distCUDA2(torch.tensor(pts).cuda()).cpu().numpy()
    want = lo.knn3_mean_dist2(pts)
    assert got.shape == (N,) and np.abs(got - want).max() <= 2e-6 * want.max() + 1e-12
    # the drop-in package exports the plug-in's symbol
    sys.path.insert(0, os.path.join(ROOT, "multiview_inpaint_amd", "dropin"))
    from simple_knn._C import distCUDA2
    assert torch.equal(distCUDA2(torch.tensor(pts).cuda()).cpu(), torch.tensor(got))


def test_distcuda2_edge_cases(T):
    assert T.distCUDA2(torch.zeros(0, 3).cuda()).shape == (0,)
    d = T.distCUDA2(torch.tensor([[0.0, 0, 0], [1.0, 0, 0]]).cuda())     # fewer than 3 neighbours: FLT_MAX terms, like the plug-in
    assert bool((d > 1e37).all())
    with pytest.raises(ValueError):
        T.distCUDA2(torch.zeros(5, 2).cuda())
    with pytest.raises(RuntimeError):
        T.distCUDA2(torch.zeros(5, 3))


# ---- prune_points / _prune_optimizer tensor surgery ------------------------------------------------

@pytest.mark.parametrize("P,frac", [(1, 1.0), (1, 0.0), (1023, 0.5), (1025, 0.9), (70001, 0.37), (4096, 0.0), (4096, 1.0)])
def test_compact_rows_equals_boolean_indexing(P, frac):
    from multiview_inpaint_amd import train_ops as T
    g = torch.Generator().manual_seed(P)
    mask = (torch.rand(P, generator=g) < frac).to(DEV)
    ts = [torch.randn(P, 3, generator=g), torch.randn(P, 1, 3, generator=g), torch.randn(P, 15, 3, generator=g),
          torch.randn(P, 1, generator=g), torch.randn(P, 4, generator=g), torch.randn(P, generator=g),
          torch.randint(0, 1 << 30, (P, 2), generator=g, dtype=torch.int32)]
    ts = [t.to(DEV) for t in ts]
    outs = T.compact_rows(mask, ts)
    for t, o in zip(ts, outs):
        assert o.dtype == t.dtype and torch.equal(o, t[mask])


def test_compact_rows_many_tensors_and_bad_arguments():
    from multiview_inpaint_amd import train_ops as T
    P = 5000
    g = torch.Generator().manual_seed(3)
    mask = (torch.rand(P, generator=g) < 0.6).to(DEV)
    ts = [torch.randn(P, 1 + (i % 5), generator=g).to(DEV) for i in range(30)]       # more than one table
    for t, o in zip(ts, T.compact_rows(mask, ts)):
        assert torch.equal(o, t[mask])
    with pytest.raises(TypeError):
        T.compact_rows(mask, [torch.zeros(P, 2, dtype=torch.float64, device=DEV)])
    with pytest.raises(ValueError):
        T.compact_rows(mask, [torch.zeros(P + 1, 2, device=DEV)])
    with pytest.raises(RuntimeError):
        T.compact_rows(mask.cpu(), [])


def test_prune_optimizer_state_matches_the_reference_recipe():
    """gaussian_model.py:351-382 on torch.optim.Adam: after pruning, parameters, both moments and the statistics equal
    the reference's per-tensor boolean indexing, and the optimizer keeps stepping."""
    from multiview_inpaint_amd import train_ops as T
    P = 3000
    g = torch.Generator().manual_seed(5)
    shapes = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (15, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,)}

    def make():
        gg = torch.Generator().manual_seed(6)
        ps = {k: torch.nn.Parameter(torch.randn(P, *s, generator=gg).to(DEV)) for k, s in shapes.items()}
        opt = torch.optim.Adam([{"params": [p], "lr": 1e-3, "name": k} for k, p in ps.items()], lr=0.0, eps=1e-15)
        for p in ps.values():
            p.grad = torch.randn(p.shape, generator=gg).to(DEV)
        opt.step()
        return ps, opt
    mask = (torch.rand(P, generator=g) < 0.7).to(DEV)
    stats = [torch.randn(P, 1, generator=g).to(DEV), torch.randn(P, generator=g).to(DEV)]
    ps_a, opt_a = make()
    new, extras = T.prune_optimizer_state(opt_a, mask, extra=stats)
    ps_b, opt_b = make()
    for grp in opt_b.param_groups:                       # the reference's loop, verbatim semantics
        stt = opt_b.state.get(grp["params"][0], None)
        stt["exp_avg"], stt["exp_avg_sq"] = stt["exp_avg"][mask], stt["exp_avg_sq"][mask]
        del opt_b.state[grp["params"][0]]
        grp["params"][0] = torch.nn.Parameter(grp["params"][0][mask].requires_grad_(True))
        opt_b.state[grp["params"][0]] = stt
    for ga, gb in zip(opt_a.param_groups, opt_b.param_groups):
        pa, pb = ga["params"][0], gb["params"][0]
        assert ga["name"] == gb["name"] and new[ga["name"]] is pa and pa.requires_grad
        assert torch.equal(pa, pb)
        assert torch.equal(opt_a.state[pa]["exp_avg"], opt_b.state[pb]["exp_avg"])
        assert torch.equal(opt_a.state[pa]["exp_avg_sq"], opt_b.state[pb]["exp_avg_sq"])
        assert opt_a.state[pa]["step"] == opt_b.state[pb]["step"]
    assert torch.equal(extras[0], stats[0][mask]) and torch.equal(extras[1], stats[1][mask])
    for ga, gb in zip(opt_a.param_groups, opt_b.param_groups):
        ga["params"][0].grad = torch.ones_like(ga["params"][0])
        gb["params"][0].grad = torch.ones_like(gb["params"][0])
    opt_a.step(); opt_b.step()
    for ga, gb in zip(opt_a.param_groups, opt_b.param_groups):
        assert torch.equal(ga["params"][0], gb["params"][0])


def test_sharded_adam_on_one_rank_equals_fused_adam(T):
    """dist.ShardedAdam with its production inner optimizer (FusedAdam, mvi_adam_step) on ONE rank over RCCL: the
    reduce-scatter / all-gather are identities there, so parameters and moments must equal plain FusedAdam bit for bit,
    body and replicated tail rows (P = 10 000: 9 984 + 16), through a learning-rate change; then the factored SH form
    against the dense one (SH gradient rebuilt by mvi_raster_sh_backward_views from the colour factor)."""
    import os
    import socket
    import torch.distributed as td
    from multiview_inpaint_amd import dist as md
    P, M, deg = 10_000, 16, 3
    shapes = {"xyz": (P, 3), "f_dc": (P, 1, 3), "f_rest": (P, M - 1, 3), "opacity": (P, 1), "scaling": (P, 3), "rotation": (P, 4)}
    lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}
    g = torch.Generator("cuda").manual_seed(0)
    init = {n: torch.randn(s, device="cuda", generator=g) for n, s in shapes.items()}
    init["xyz"] = init["xyz"] + torch.tensor([0.0, 0.0, 5.0], device="cuda")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("nccl", rank=0, world_size=1)
    try:
        mine = {n: t.clone() for n, t in init.items()}
        opt = md.ShardedAdam(mine, lrs, eps=1e-15)
        assert isinstance(opt.inner, T.FusedAdam) and (opt.plan.rows, opt.plan.tail) == (9984, 16)
        ref_p = {n: torch.nn.Parameter(t.clone()) for n, t in init.items()}
        ref = T.FusedAdam([{"params": [p], "lr": lrs[n], "name": n} for n, p in ref_p.items()], lr=0.0, eps=1e-15)
        for it in range(3):
            grads = {n: torch.randn(sh, device="cuda", generator=g) for n, sh in shapes.items()}
            if it == 2:
                opt.set_lr("xyz", 7e-5)
                [q for q in ref.param_groups if q["name"] == "xyz"][0]["lr"] = 7e-5
            opt.step(grads)
            for n, p in ref_p.items():
                p.grad = grads[n].clone()
            ref.step()
            for n in shapes:
                assert torch.equal(mine[n], ref_p[n].data), (it, n)
        full = opt.full_state()
        for n, p in ref_p.items():
            assert torch.equal(full[n]["exp_avg"], ref.state[p]["exp_avg"]) and torch.equal(full[n]["exp_avg_sq"], ref.state[p]["exp_avg_sq"]), n
        # factored SH gradient == the dense gradient it stands for
        a = {n: t.clone() for n, t in init.items()}
        b = {n: t.clone() for n, t in init.items()}
        oa, ob = md.ShardedAdam(a, lrs, eps=1e-15), md.ShardedAdam(b, lrs, eps=1e-15)
        cam = torch.tensor([0.3, -0.2, 0.1], device="cuda")
        fac = torch.randn(P, 3, device="cuda", generator=g)
        small = {n: torch.randn(shapes[n], device="cuda", generator=g) for n in ("xyz", "opacity", "scaling", "rotation")}
        sh = md.sh_grad_from_factors(init["xyz"], cam[None], fac[None], M, deg)
        oa.step(dict(small, f_dc=sh[:, :1].contiguous(), f_rest=sh[:, 1:].contiguous()))
        ob.step_factored(small, fac, cam, init["xyz"], deg)
        torch.cuda.synchronize()
        for n in shapes:
            assert torch.equal(a[n], b[n]), n
    finally:
        td.destroy_process_group()
