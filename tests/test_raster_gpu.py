"""GPU parity tests of the HIP rasterizer (through the C-ABI) against the CPU oracle.

Bars (BASELINE.json north_star): bit-exact on integer outputs (radii, tiles touched, sort keys,
sorted order, tile ranges) — and, because the per-Gaussian arithmetic contract is shared, bit-exact
on the per-Gaussian floats too; 1e-4 relative on RGB-D and on every gradient.

The compositing loop takes three hard decisions per (pixel, Gaussian) pair (power > 0,
alpha < 1/255, T(1-alpha) < 1e-4) and one for the median depth (T crossing 0.5). exp() differs by
~1 ulp between v_exp_f32 and glibc, so a handful of pairs per 10^8 can take the other branch; such a
pixel then differs by up to alpha*T*c <= 4e-3 in colour (or picks the neighbouring Gaussian's depth).
FLIP_FRAC bounds the fraction of pixels allowed outside the 1e-4 band for that reason; every other
pixel must be inside it."""
import numpy as np
import pytest
import torch

from multiview_inpaint_amd import synthetic as syn
from raster_helpers import oracle_params, small_scene

pytestmark = pytest.mark.gpu
RTOL = 1e-4
FLIP_FRAC = 2e-4


@pytest.fixture(scope="module")
def R():
    from multiview_inpaint_amd import raster
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return raster


@pytest.fixture(scope="module")
def ro():
    from oracle import raster_oracle
    return raster_oracle


def _settings(R, cam, bg, deg, scale_modifier=1.0):
    d = "cuda"
    return R.GaussianRasterizationSettings(
        image_height=cam["H"], image_width=cam["W"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
        bg=torch.tensor(bg, device=d), scale_modifier=scale_modifier,
        viewmatrix=torch.tensor(cam["viewmatrix"], device=d), projmatrix=torch.tensor(cam["projmatrix"], device=d),
        sh_degree=deg, campos=torch.tensor(cam["campos"], device=d), prefiltered=False)


def _to_dev(sc):
    return {k: torch.tensor(v, device="cuda") for k, v in sc.items() if k != "sh_degree"}


def _close_frac(a, b, rtol=RTOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = np.abs(b).max() + 1e-12
    bad = np.abs(a - b) > rtol * scale
    return bad.mean(), np.abs(a - b).max() / scale


def _check_forward(R, ro, cam, sc, bg, full=True):
    deg = sc["sh_degree"]
    p = oracle_params(ro, cam, sc, bg)
    f = ro.forward(p, sc["means3D"], sc["opacities"], shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    t = _to_dev(sc)
    rs = _settings(R, cam, bg, deg)
    color, radii, depth, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"],
                                                  rotations=t["rotations"])
    torch.cuda.synchronize()
    P, D, W, H = st.P, st.D, cam["W"], cam["H"]
    # ---- integer outputs: bit-exact
    assert D == f["num_rendered"]
    assert np.array_equal(radii.cpu().numpy(), f["radii"])
    vis = f["radii"] > 0
    tt = st.tensor("tiles_touched", (P,), torch.int32).cpu().numpy().view(np.uint32)
    assert np.array_equal(tt, f["tiles_touched"])
    plist = st.tensor("point_list", (D,), torch.int32).cpu().numpy().view(np.uint32)
    assert np.array_equal(plist, f["point_list"]), "sorted order differs"
    # the 64-bit sort key of pair i is tile_ids_sorted[i] << 32 | bits(depths[point_list[i]]) (include/mvi_raster.h)
    tids = st.tensor("tile_ids_sorted", (D,), torch.int32).cpu().numpy().view(np.uint32)
    dbits = st.tensor("depths", (P,), torch.float32).cpu().numpy().view(np.uint32)
    keys = (tids.astype(np.uint64) << np.uint64(32)) | dbits[plist].astype(np.uint64)
    assert np.array_equal(keys, f["keys_sorted"]), "sort keys differ"
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    ranges = st.tensor("ranges", (tiles, 2), torch.int32).cpu().numpy().view(np.uint32)
    assert np.array_equal(ranges, f["ranges"])
    # ---- per-Gaussian floats: shared arithmetic contract -> bit-exact where visible
    cov = np.concatenate([st.tensor("cov3D_a", (P, 4), torch.float32).cpu().numpy(),
                          st.tensor("cov3D_b", (P, 2), torch.float32).cpu().numpy()], 1)
    rgbd = st.tensor("rgbd", (P, 4), torch.float32).cpu().numpy()
    got = dict(depths=st.tensor("depths", (P,), torch.float32).cpu().numpy(),
               means2D=st.tensor("means2D", (P, 2), torch.float32).cpu().numpy(), cov3D=cov,
               conic_opacity=st.tensor("conic_opacity", (P, 4), torch.float32).cpu().numpy(), rgb=rgbd[:, :3])
    for name, ref in (("depths", f["depths"]), ("means2D", f["xy"]), ("cov3D", f["cov3D"]),
                      ("conic_opacity", f["conic_opacity"]), ("rgb", f["rgb"])):
        assert np.array_equal(got[name][vis], ref[vis]), name
    assert np.array_equal(rgbd[vis, 3], f["depths"][vis])
    cl = st.tensor("clamped", (P,), torch.uint8).cpu().numpy()
    want = (f["clamped"][:, 0] | (f["clamped"][:, 1] << 1) | (f["clamped"][:, 2] << 2)).astype(np.uint8)
    assert np.array_equal(cl[vis], want[vis])
    # ---- image outputs: 1e-4 rel outside threshold flips
    frac, worst = _close_frac(color.cpu().numpy(), f["color"])
    assert frac <= FLIP_FRAC and worst < 2e-2, (frac, worst)
    nc = st.tensor("n_contrib", (H, W), torch.int32).cpu().numpy().view(np.uint32)
    assert (nc != f["n_contrib"]).mean() <= FLIP_FRAC
    dfrac, _ = _close_frac(depth.cpu().numpy(), f["depth"])
    assert dfrac <= FLIP_FRAC
    assert (depth.cpu().numpy()[0][f["n_contrib"] == 0] == 15.0).all()
    ft = st.tensor("final_T", (H, W), torch.float32).cpu().numpy()
    tfrac, _ = _close_frac(ft, f["final_T"])
    assert tfrac <= FLIP_FRAC
    return f, p, rs, t, st


@pytest.mark.parametrize("seed,deg,pose,N,W,H", [(0, 3, True, 400, 100, 70), (1, 0, True, 3000, 200, 120),
                                                 (2, 1, False, 2000, 129, 67), (3, 2, True, 5000, 320, 240)])
def test_forward_parity_small(R, ro, seed, deg, pose, N, W, H):
    cam, sc, bg = small_scene(seed, N=N, W=W, H=H, deg=deg, pose=pose, log_scale=np.log(0.04))
    _check_forward(R, ro, cam, sc, bg)


@pytest.mark.parametrize("N", [1, 63, 64, 65, 127, 129, 257, 4097])
def test_forward_parity_block_boundaries_and_large_footprints(R, ro, N):
    """Gaussian counts around the kernels' block sizes (64-thread preprocess blocks, 256-entry emission blocks,
    2048 / 4096-pair sort tiles) with LARGE footprints — every Gaussian covers many tiles, one Gaussian's pairs span
    emission rounds and sort tiles: binning integers bit-exact, image within tolerance."""
    cam, sc, bg = small_scene(100 + N, N=N, W=176, H=112, deg=1, pose=True, log_scale=np.log(0.6))
    _check_forward(R, ro, cam, sc, bg)


def test_parity_with_more_than_65536_tiles(R, ro):
    """4112 x 4096 pixels = 257 x 256 tiles: tile ids no longer fit the 16-bit sort keys the binning uses up to 65536
    tiles, so this runs its 32-bit instantiation (emission, both radix passes, tile ranges) — integers bit-exact, image
    and gradients within tolerance, and pairs with tile ids above 65535 must exist for the case to mean anything."""
    cam, sc, bg = small_scene(77, N=600, W=4112, H=4096, deg=1, pose=True, log_scale=np.log(0.02))
    f, p, rs, t, st = _check_forward(R, ro, cam, sc, bg)
    assert st.views().tile_id_bytes == 4
    assert ((f["keys_sorted"] >> np.uint64(32)) > 65535).sum() > 100
    g_img = np.random.default_rng(77).normal(size=(3, cam["H"], cam["W"])).astype(np.float32)
    okw = dict(shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    b = ro.backward(p, f, g_img, sc["means3D"], **okw)
    g = R.rasterize_backward(rs, st, torch.tensor(g_img, device="cuda"), t["means3D"], shs=t["shs"], scales=t["scales"],
                             rotations=t["rotations"])
    torch.cuda.synchronize()
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations"):
        frac, worst = _close_frac(g[k].cpu().numpy(), b[k])
        assert frac <= 5e-4 and worst < 5e-2, (k, frac, worst)


@pytest.mark.parametrize("seed,deg,mode", [(0, 3, "sh"), (4, 1, "sh"), (5, 2, "precomp"), (6, 0, "sh")])
def test_backward_parity_small(R, ro, seed, deg, mode):
    cam, sc, bg = small_scene(seed, N=1500, W=160, H=112, deg=deg, log_scale=np.log(0.05))
    t = _to_dev(sc)
    p = oracle_params(ro, cam, sc, bg)
    rs = _settings(R, cam, bg, deg)
    g_img = np.random.default_rng(seed).normal(size=(3, cam["H"], cam["W"])).astype(np.float32)
    if mode == "precomp":
        f0 = ro.forward(p, sc["means3D"], sc["opacities"], shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"], render=False)
        c6, cp = f0["cov3D"].copy(), np.abs(np.random.default_rng(9).normal(0.5, 0.3, (p.P, 3))).astype(np.float32)
        okw = dict(colors_precomp=cp, cov3D_precomp=c6)
        gkw = dict(colors_precomp=torch.tensor(cp, device="cuda"), cov3D_precomp=torch.tensor(c6, device="cuda"))
    else:
        okw = dict(shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
        gkw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    f = ro.forward(p, sc["means3D"], sc["opacities"], **okw)
    b = ro.backward(p, f, g_img, sc["means3D"], **okw)
    color, radii, depth, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], **gkw)
    g = R.rasterize_backward(rs, st, torch.tensor(g_img, device="cuda"), t["means3D"], **gkw)
    torch.cuda.synchronize()
    for k in ("means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"):
        if b[k] is None:
            assert g[k] is None, k
            continue
        frac, worst = _close_frac(g[k].cpu().numpy(), b[k])
        assert frac <= 5e-4 and worst < 5e-2, (k, frac, worst)


def test_autograd_module_contract(R, ro):
    """The nn.Module surface the reference calls (gaussian_renderer/__init__.py:85-101)."""
    cam, sc, bg = small_scene(7, N=800, W=96, H=64, deg=1, log_scale=np.log(0.06))
    t = {k: v.requires_grad_(True) for k, v in _to_dev(sc).items()}
    rs = _settings(R, cam, bg, 1)
    screen = torch.zeros_like(t["means3D"], requires_grad=True) + 0
    screen.retain_grad()
    rast = R.GaussianRasterizer(raster_settings=rs)
    out = rast(means3D=t["means3D"], means2D=screen, shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
               scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    assert isinstance(out, tuple) and len(out) == 3
    color, radii, depth = out
    assert color.shape == (3, 64, 96) and depth.shape == (1, 64, 96) and radii.shape == (800,) and radii.dtype == torch.int32
    assert not depth.requires_grad and color.requires_grad
    target = torch.rand_like(color)
    (color - target).abs().mean().backward()
    assert screen.grad is not None and screen.grad.shape == (800, 3) and (screen.grad[:, 2] == 0).all()
    vis = radii > 0
    assert (screen.grad[~vis] == 0).all() and screen.grad[vis].abs().sum() > 0
    for k in ("means3D", "shs", "opacities", "scales", "rotations"):
        assert t[k].grad is not None and t[k].grad.shape == t[k].shape and torch.isfinite(t[k].grad).all()
    # no-grad forward (gen_seq.py:65) and the argument-combination errors of the plug-in
    with torch.no_grad():
        c2, r2, d2 = rast(means3D=t["means3D"], means2D=screen, shs=t["shs"], opacities=t["opacities"],
                          scales=t["scales"], rotations=t["rotations"])
    assert torch.equal(c2, color.detach())
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(means3D=t["means3D"], means2D=screen, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=t["means3D"], means2D=screen, shs=t["shs"], opacities=t["opacities"])
    assert torch.equal(rast.markVisible(t["means3D"].detach()).cpu(),
                       torch.tensor((sc["means3D"] @ cam["viewmatrix"][:3, 2] + cam["viewmatrix"][3, 2]) > 0.2))


def test_empty_and_culled(R):
    cam = syn.make_camera(48, 32, 60.0)
    bg = np.array([0.2, 0.4, 0.6], np.float32)
    rs = _settings(R, cam, bg, 0)
    rast = R.GaussianRasterizer(rs)
    N = 16
    means = torch.zeros(N, 3, device="cuda")
    means[:, 2] = torch.linspace(-3, 0.2, N)
    kw = dict(means2D=torch.zeros(N, 3, device="cuda"), shs=torch.ones(N, 1, 3, device="cuda"),
              opacities=torch.full((N, 1), 0.9, device="cuda"), scales=torch.full((N, 3), 0.1, device="cuda"),
              rotations=torch.tensor([[1.0, 0, 0, 0]], device="cuda").repeat(N, 1))
    means.requires_grad_(True)
    color, radii, depth = rast(means3D=means, **kw)
    assert (radii == 0).all() and (depth == 15.0).all()
    assert torch.allclose(color, torch.tensor(bg, device="cuda")[:, None, None].expand_as(color))
    color.sum().backward()
    assert (means.grad == 0).all()
    # P = 0
    z = lambda *s: torch.zeros(*s, device="cuda")
    color, radii, depth = rast(means3D=z(0, 3), means2D=z(0, 3), shs=z(0, 1, 3), opacities=z(0, 1), scales=z(0, 3), rotations=z(0, 4))
    assert radii.numel() == 0 and (depth == 15.0).all()
    assert torch.allclose(color, torch.tensor(bg, device="cuda")[:, None, None].expand_as(color))


def test_bringup_config_100k_800(R, ro):
    """BASELINE.json configs[1]: 100k synthetic Gaussians, 800x800, fwd+bwd vs the oracle."""
    cam = syn.make_camera(800, 800, 50.0)
    sc = syn.make_scene(100_000, cam, 3, seed=0)
    bg = np.zeros(3, np.float32)
    f, p, rs, t, st = _check_forward(R, ro, cam, sc, bg)
    g_img = np.random.default_rng(0).normal(size=(3, 800, 800)).astype(np.float32)
    b = ro.backward(p, f, g_img, sc["means3D"], shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    g = R.rasterize_backward(rs, st, torch.tensor(g_img, device="cuda"), t["means3D"], shs=t["shs"], scales=t["scales"],
                             rotations=t["rotations"])
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations"):
        frac, worst = _close_frac(g[k].cpu().numpy(), b[k])
        assert frac <= 5e-4, (k, frac, worst)


def test_full_size_properties_1p5M_1080p(R):
    """BASELINE.json configs[2] size (1.5M Gaussians, 1920x1080): size-independent properties."""
    cam = syn.make_camera(1920, 1080, 50.0)
    sc = syn.make_scene(1_500_000, cam, 3, seed=0)
    t = _to_dev(sc)
    W, H = 1920, 1080
    kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    bg0, bg1 = np.zeros(3, np.float32), np.array([0.25, 0.5, 1.0], np.float32)
    c0, radii, d0, st = R.rasterize_forward(_settings(R, cam, bg0, 3), t["means3D"], t["opacities"], **kw)
    P, D = st.P, st.D
    plist = st.tensor("point_list", (D,), torch.int32).long()
    tt = st.tensor("tiles_touched", (P,), torch.int32).long()
    depths = st.tensor("depths", (P,), torch.float32)
    tids = st.tensor("tile_ids_sorted", (D,), torch.int32).long()
    keys = (tids << 32) | (depths[plist].view(torch.int32).long() & 0xFFFFFFFF)
    # sortedness (keys are non-negative as int64: tile < 2^31) and key <-> payload consistency
    assert (keys[1:] >= keys[:-1]).all()
    assert D == int(tt.sum()) and ((radii > 0) == (tt > 0)).all()
    assert torch.equal((keys & 0xFFFFFFFF).int(), depths[plist].view(torch.int32))
    # the sorted payload is a permutation of "Gaussian i repeated tiles_touched[i] times"
    assert torch.equal(torch.bincount(plist, minlength=P), tt)
    # stability: equal keys keep ascending Gaussian index
    same = keys[1:] == keys[:-1]
    assert (plist[1:][same] > plist[:-1][same]).all()
    # ranges partition [0, D) in tile order
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    ranges = st.tensor("ranges", (tiles, 2), torch.int32).long()
    tile_of = keys >> 32
    cnt = torch.bincount(tile_of, minlength=tiles)
    assert torch.equal(ranges[:, 1] - ranges[:, 0], cnt)
    nz = cnt > 0
    assert torch.equal(ranges[nz][:, 0], (torch.cumsum(cnt, 0) - cnt)[nz])
    # background linearity: color(bg) - color(0) == final_T * bg ; depth independent of bg
    c1, _, d1, st1 = R.rasterize_forward(_settings(R, cam, bg1, 3), t["means3D"], t["opacities"], **kw)
    ft = st1.tensor("final_T", (H, W), torch.float32)
    assert torch.allclose(c1 - c0, ft[None] * torch.tensor(bg1, device="cuda")[:, None, None], atol=1e-6)
    assert torch.equal(d0, d1)
    assert ((d0 == 15.0) | ((d0 > 0.2) & (d0 < 9.0))).all()
    # gradient linearity in the upstream gradient (atomics reorder sums: tolerance, not bit-exact)
    g_img = torch.randn(3, H, W, device="cuda")
    ga = R.rasterize_backward(_settings(R, cam, bg0, 3), st, g_img, t["means3D"], **kw)
    gb = R.rasterize_backward(_settings(R, cam, bg0, 3), st, 2 * g_img, t["means3D"], **kw)
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations"):
        assert torch.isfinite(ga[k]).all()
        err = (gb[k] - 2 * ga[k]).abs().max() / (gb[k].abs().max() + 1e-12)
        assert err < 1e-4, (k, float(err))
        assert (ga[k][radii == 0] == 0).all()


def test_partial_sh_degree_scale_modifier_and_background_gradient(R, ro):
    """sh_degree below the stored coefficient count (the reference raises the active degree during training,
    gs-simp/scene/gaussian_model.py:119-121), scale_modifier != 1 and a non-zero background: forward + backward."""
    cam, sc, bg = small_scene(8, N=1200, W=144, H=96, deg=3, log_scale=np.log(0.05))
    sc = dict(sc, sh_degree=1)                             # 16 coefficients stored, degree 1 active
    t = _to_dev(sc)
    p = oracle_params(ro, cam, sc, bg, scale_modifier=0.7)
    rs = _settings(R, cam, bg, 1, scale_modifier=0.7)
    kw = dict(shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    f = ro.forward(p, sc["means3D"], sc["opacities"], **kw)
    g_img = np.random.default_rng(8).normal(size=(3, 96, 144)).astype(np.float32)
    b = ro.backward(p, f, g_img, sc["means3D"], **kw)
    color, radii, depth, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"],
                                                  rotations=t["rotations"])
    g = R.rasterize_backward(rs, st, torch.tensor(g_img, device="cuda"), t["means3D"], shs=t["shs"], scales=t["scales"],
                             rotations=t["rotations"])
    assert np.array_equal(radii.cpu().numpy(), f["radii"])
    frac, worst = _close_frac(color.cpu().numpy(), f["color"])
    assert frac <= FLIP_FRAC, (frac, worst)
    assert (g["shs"][:, 4:] == 0).all()                   # inactive coefficients get exactly zero gradient
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations"):
        frac, worst = _close_frac(g[k].cpu().numpy(), b[k])
        assert frac <= 5e-4 and worst < 5e-2, (k, frac, worst)


@pytest.mark.parametrize("deg,store_deg", [(3, 3), (1, 3), (0, 0)])
def test_factored_sh_gradient_and_multi_view_rebuild(R, deg, store_deg):
    """View-parallel exchange form of the SH gradient: the colour factor of each view
    (sh_grad="factor") plus mvi_raster_sh_backward_views reproduces the SUM of the views' dense SH gradients
    (dense gradients are oracle-checked in test_backward_parity_small)."""
    from multiview_inpaint_amd import synthetic as syn
    rng = np.random.default_rng(11)
    Rm, T0 = syn.random_rotation(rng), rng.normal(size=3)
    cams = [syn.make_camera(160, 112, 50.0, Rm, T0), syn.make_camera(160, 112, 50.0, Rm, T0 + np.array([0.35, -0.2, 0.1]))]
    sc = syn.make_scene(1500, cams[0], store_deg, 11, log_scale_mean=np.log(0.05), zmin=1.0, zmax=6.0)
    bg = np.array([0.3, 0.1, 0.7], np.float32)
    t = _to_dev(sc)
    P, M = t["shs"].shape[0], t["shs"].shape[1]
    kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    dense, factors, campos = [], [], []
    for v, cam in enumerate(cams):
        rs = _settings(R, cam, bg, deg)
        g_img = torch.tensor(np.random.default_rng(20 + v).normal(size=(3, cam["H"], cam["W"])).astype(np.float32), device="cuda")
        _, radii, _, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], **kw)
        assert int((radii > 0).sum()) > P // 3
        gd = R.rasterize_backward(rs, st, g_img, t["means3D"], **kw)
        gb = R.rasterize_backward(rs, st, g_img, t["means3D"], sh_grad="both", **kw)
        gf = R.rasterize_backward(rs, st, g_img, t["means3D"], sh_grad="factor", **kw)
        assert gf["shs"] is None and gf["sh_color_factor"].shape == (P, 3)
        for k in ("means3D", "opacities", "scales", "rotations"):
            frac, worst = _close_frac(gf[k].cpu().numpy(), gd[k].cpu().numpy())
            assert frac <= 5e-4 and worst < 5e-2, (k, frac, worst)          # atomics: summation order only
        frac, worst = _close_frac(gf["sh_color_factor"].cpu().numpy(), gb["sh_color_factor"].cpu().numpy())
        assert frac <= 5e-4 and worst < 5e-2
        # the degree-0 coefficient of the dense gradient is C0 * factor
        frac, worst = _close_frac((gb["sh_color_factor"] * 0.28209479177387814).cpu().numpy(), gb["shs"][:, 0].cpu().numpy())
        assert frac <= 1e-5 and worst < 1e-5
        dense.append(gb["shs"])
        factors.append(gb["sh_color_factor"])
        campos.append(rs.campos)
    both, cp = torch.stack(factors), torch.stack(campos)
    for n_views in (1, 2):
        out = R.sh_backward_views(t["means3D"], cp[:n_views], both[:n_views], M, deg)
        want = sum(dense[:n_views])
        frac, worst = _close_frac(out.cpu().numpy(), want.cpu().numpy(), rtol=1e-5)
        assert out.shape == (P, M, 3) and frac == 0.0, (n_views, frac, worst)
    # strided inputs, as they sit in the all-gathered buffer [W, 3P + 3]
    packed = torch.zeros(2, 3 * P + 3, device="cuda")
    packed[:, :3 * P] = both.reshape(2, -1)
    packed[:, 3 * P:] = cp
    out2 = R.sh_backward_views(t["means3D"], packed[:, 3 * P:], packed[:, :3 * P].view(2, P, 3), M, deg)
    assert torch.equal(out2, R.sh_backward_views(t["means3D"], cp, both, M, deg))
    with pytest.raises(ValueError):
        R.sh_backward_views(t["means3D"], cp[:1], both, M, deg)


@pytest.mark.parametrize("deg,store_deg", [(3, 3), (1, 2), (0, 0)])
def test_raw_parameter_path_equals_activations_plus_standard_path(R, deg, store_deg):
    """forward_raw (sigmoid / exp / normalize / SH concat inside the preprocess kernels, chain rule in the backward)
    against train_ops.activate_gaussians + the standard path: same integer outputs bit for bit (both use the same
    device expressions for the activations), images and raw-parameter gradients to summation order."""
    from multiview_inpaint_amd import train_ops as T
    cam, sc, bg = small_scene(21, N=1500, W=160, H=112, deg=store_deg, log_scale=np.log(0.05))
    t = _to_dev(sc)
    P, M = t["shs"].shape[0], t["shs"].shape[1]
    raw = dict(xyz=t["means3D"], dc=t["shs"][:, :1].contiguous(), rest=t["shs"][:, 1:].contiguous(),
               o=torch.logit(t["opacities"].clamp(1e-4, 1 - 1e-4)), s=torch.log(t["scales"]),
               q=t["rotations"] * (0.5 + torch.rand(P, 1, device="cuda", generator=torch.Generator("cuda").manual_seed(1))))
    rs = _settings(R, cam, bg, deg)
    g_img = torch.tensor(np.random.default_rng(5).normal(size=(3, cam["H"], cam["W"])).astype(np.float32), device="cuda")
    rast = R.GaussianRasterizer(rs)

    def run(raw_mode):
        p = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
        m2d = torch.zeros(P, 3, device="cuda", requires_grad=True)
        if raw_mode:
            color, radii, depth = rast.forward_raw(p["xyz"], m2d, p["dc"], p["rest"], p["o"], p["s"], p["q"])
        else:
            scales, rots, opac, shs = T.activate_gaussians(p["s"], p["q"], p["o"], p["dc"], p["rest"])
            color, radii, depth = rast(means3D=p["xyz"], means2D=m2d, shs=shs, opacities=opac, scales=scales, rotations=rots)
        (color * g_img).sum().backward()
        return color.detach(), radii, depth, {**{k: v.grad for k, v in p.items()}, "m2d": m2d.grad}

    c0, r0, d0, g0 = run(False)
    c1, r1, d1, g1 = run(True)
    assert torch.equal(r0, r1) and int((r0 > 0).sum()) > P // 3
    assert torch.equal(c0, c1) and torch.equal(d0, d1)                 # forward has no atomics: identical
    for k in g0:
        assert g1[k] is not None and g1[k].shape == g0[k].shape, k
        if g0[k].numel() == 0:
            continue                                                   # degree 0: features_rest is [P,0,3]
        frac, worst = _close_frac(g1[k].cpu().numpy(), g0[k].cpu().numpy())
        assert frac <= 5e-4 and worst < 5e-2, (k, frac, worst)


def test_split_backward_equals_one_call_and_factors_come_early(R):
    """mvi_raster_backward_render + mvi_raster_backward_geom == mvi_raster_backward (to the order of the float atomics), and
    the colour factors are final when the `after_render` hook runs (what dist.FactoredGradExchange.begin_gather relies on)."""
    cam, sc, bg = small_scene(7, N=3000, W=200, H=120, deg=3, pose=True, log_scale=np.log(0.05))
    t = _to_dev(sc)
    rs = _settings(R, cam, bg, 3)
    color, radii, depth, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"],
                                                  rotations=t["rotations"])
    g_img = torch.randn(3, cam["H"], cam["W"], device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    one = R.rasterize_backward(rs, st, g_img, t["means3D"], sh_grad="factor", **kw)
    P = t["means3D"].shape[0]
    z = lambda *s: torch.full(s, float("nan"), device="cuda")
    out = dict(means3D=z(P, 3), means2D=z(P, 3), opacities=z(P, 1), scales=z(P, 3), rotations=z(P, 4), sh_color_factor=z(P, 3))
    seen = {}

    def hook():
        torch.cuda.synchronize()
        seen["factors"] = out["sh_color_factor"].clone()
        seen["others_untouched"] = bool(torch.isnan(out["means3D"]).all())
    two = R.rasterize_backward_split(rs, st, g_img, t["means3D"], t["shs"], t["scales"], t["rotations"], out, after_render=hook)
    torch.cuda.synchronize()
    # the factors were final at the hook (the chain-rule kernel does not touch that buffer afterwards)
    assert seen["others_untouched"] and torch.equal(seen["factors"], two["sh_color_factor"])
    # two runs of the render backward differ in the order of their float atomics: equal to summation order
    for k in ("means3D", "means2D", "opacities", "scales", "rotations", "sh_color_factor"):
        a, b = two[k].double(), one[k].double()
        assert float((a - b).abs().max() / (b.abs().max() + 1e-30)) < 1e-5, k
