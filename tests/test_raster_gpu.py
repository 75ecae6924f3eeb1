"""GPU parity tests of the HIP rasterizer (through the C-ABI) against the CPU oracle.

Bars (BASELINE.json north_star): bit-exact on integer outputs (radii, tiles touched, sort keys,
sorted order, tile ranges) — and, because the per-Gaussian arithmetic contract is shared, bit-exact
on the per-Gaussian floats too; 1e-4 relative, ELEMENTWISE, on RGB-D and on every gradient:
|got - ref| <= 1e-4 * max(|ref|, floor), with the floor stated at each use.

The compositing loop takes three hard decisions per (pixel, Gaussian) pair (power > 0, alpha < 1/255,
T(1-alpha) < 1e-4) and one for the median depth (T crossing 0.5). exp() differs by ~1 ulp between
v_exp_f32 and glibc, so a pair whose tested quantity sits within rounding distance of its threshold
can take the other branch. The oracle says which pixels hold such a pair (raster_oracle.margins, relative
margin 2e-5): every OTHER pixel must meet the full tolerance (depth and n_contrib: exactly), and the
marginal pixels that differ are COUNTED against a small allowance (expected flips ~ 2e-8 per evaluated
pair: 0 at the small test sizes, a few at 100k Gaussians / 800x800)."""
import os

import numpy as np
import pytest
import torch

from multiview_inpaint_amd import synthetic as syn
from raster_helpers import oracle_params, small_scene

pytestmark = pytest.mark.gpu
RTOL = 1e-4
GRAD_FLOOR = 1e-2          # gradient floor as a fraction of the array's largest |value|: float atomics reorder sums of
                           # hundreds of terms, so an element that cancels to ~0 is held to 1e-6 of the largest element


def _flip_allowance(n_pairs):
    """Pixels allowed to differ because a decision flipped: 2 + 1e-7 per evaluated (pixel, Gaussian) pair."""
    return 2 + int(1e-7 * n_pairs)


def _viol(got, ref, floor):
    """Elementwise bar: boolean array of violations of |got - ref| <= RTOL * max(|ref|, floor), and the worst ratio."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    tol = RTOL * np.maximum(np.abs(ref), floor)
    d = np.abs(got - ref)
    return d > tol, (float((d / tol).max()) if d.size else 0.0)


PURE_REL_MIN_FRACTION = 0.985     # of the non-zero reference elements of every gradient array, see _grad_check (round 6: was 0.97;
                                  # observed 0.9918 - 0.9989 in every test but the one below)
PURE_REL_MIN_FRACTION_HUGE_SPLATS = 0.97   # test_parity_with_more_than_65536_tiles: 600 Gaussians that each cover thousands of
                                  # pixels, every element a sum of thousands of cancelling atomic terms (observed 0.9757 - 0.992)
_REPORT = os.environ.get("MVI_PARITY_REPORT")      # file the per-array parity figures are appended to (the GPU runs set it)


def _grad_check(name, got, ref, n_flips_allowed, pure_min=None):
    """A gradient array against the oracle's: rows (Gaussians) with any element outside the elementwise bar are counted
    against the flip allowance (a flipped pair changes that Gaussian's gradient and little else); nothing may be off by
    more than 1e-2 of the array's scale.
    north_star says "within 1e-4 rel" and the bar above has a floor (1e-2 of the array's largest |value|), so the figures WITHOUT
    the floor are computed, printed and asserted too: the fraction of the non-zero reference elements that meet the pure relative
    bar |got - ref| <= 1e-4 |ref|, and the quantiles of the pure relative error. An element that is the sum of hundreds of
    cancelling float-atomic terms cannot meet 1e-4 of ITSELF when it cancels to 1e-6 of its terms — that is what the floor is
    for — so the pure figure is a measured fraction, not 1."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-30
    bad, worst = _viol(got, ref, GRAD_FLOOR * scale)
    rows = bad.reshape(bad.shape[0], -1).any(1).sum() if bad.ndim > 1 else bad.sum()
    nz = ref != 0
    rel = np.abs(got - ref)[nz] / np.abs(ref[nz]) if nz.any() else np.zeros(1)
    frac_pure = float((rel <= RTOL).mean())
    q50, q99, q999 = (float(np.quantile(rel, q)) for q in (0.5, 0.99, 0.999))
    line = (f"grad {name}: shape {tuple(ref.shape)} non-zero refs {int(nz.sum())}; pure relative 1e-4 met by {frac_pure:.5f} of them; "
            f"relative error median {q50:.2e}, p99 {q99:.2e}, p99.9 {q999:.2e}, worst {float(rel.max()):.2e}; with the floor: worst "
            f"ratio to the bar {worst:.3f}, rows outside it {int(rows)} (allowed {2 * n_flips_allowed})")
    print(line)
    if _REPORT:
        with open(_REPORT, "a") as fh:
            fh.write(os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0] + " | " + line + "\n")
    assert rows <= 2 * n_flips_allowed, (name, int(rows), worst)
    assert np.abs(got - ref).max() <= 1e-2 * scale, (name, float(np.abs(got - ref).max() / scale))
    assert frac_pure >= (PURE_REL_MIN_FRACTION if pure_min is None else pure_min), (name, frac_pure)
    return worst


def _same_to_summation_order(a, b, tol=2e-5):
    """Two GPU evaluations of the same sums (float atomics in a different order): max |a - b| relative to the scale."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30)) <= tol


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


def _check_forward(R, ro, cam, sc, bg, full=True, max_marginal=5e-3):
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
    # SH colours are evaluated on first use by the render kernel (deferred colours, include/mvi_raster.h): what it evaluated
    # must already be the oracle's bits, whatever it left pending is (-1, -1, -1, depth) / clamped 0 and is evaluated now
    rgbd0 = st.tensor("rgbd", (P, 4), torch.float32).cpu().numpy()
    cl0 = st.tensor("clamped", (P,), torch.uint8).cpu().numpy()
    pending = (rgbd0[:, :3] < 0).any(1) & vis
    done = vis & ~pending
    assert np.array_equal(rgbd0[done, :3], f["rgb"][done]) and np.array_equal(rgbd0[vis, 3], f["depths"][vis])
    assert (rgbd0[pending, :3] == -1.0).all() and (cl0[pending] == 0).all()
    if D:
        nc_ = st.tensor("n_contrib", (H, W), torch.int32).cpu().numpy()
        assert done.any() or not (nc_ > 0).any(), "nothing was evaluated although pixels composited something"
    st.resolve_colors()
    rgbd = st.tensor("rgbd", (P, 4), torch.float32).cpu().numpy()
    assert np.array_equal(rgbd[done], rgbd0[done]), "resolve_colors touched a colour that was already evaluated"
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
    # ---- image outputs: elementwise 1e-4 on every pixel the oracle does not mark marginal; marginal ones are counted
    fr = ro.margins(p, f)
    colour_marginal, any_marginal = (fr & 1) != 0, fr != 0
    assert any_marginal.mean() < max_marginal, "margin too wide: the test would excuse too many pixels"
    allow = _flip_allowance(int(f["n_contrib"].sum()))
    bad_c, worst_c = _viol(color.cpu().numpy(), f["color"], 1.0)                       # floor 1.0: colours are O(1)
    bad_c = bad_c.any(0)
    assert not (bad_c & ~colour_marginal).any(), ("colour off on a pixel with no marginal decision", worst_c)
    nc = st.tensor("n_contrib", (H, W), torch.int32).cpu().numpy().view(np.uint32)
    bad_n = nc != f["n_contrib"]
    assert not (bad_n & ~colour_marginal).any(), "n_contrib differs on a pixel with no marginal decision"
    ft = st.tensor("final_T", (H, W), torch.float32).cpu().numpy()
    bad_t, worst_t = _viol(ft, f["final_T"], 1e-4)                                     # floor = the loop's own T cut-off
    assert not (bad_t & ~colour_marginal).any(), ("final_T off on a pixel with no marginal decision", worst_t)
    dg = depth.cpu().numpy()[0]
    bad_d = dg != f["depth"][0]                                                        # a selected Gaussian's depth: exact
    assert not (bad_d & ~any_marginal).any(), "median depth differs on a pixel with no marginal decision"
    flips = int((bad_c | bad_n | bad_t | bad_d).sum())
    assert flips <= allow, (flips, allow)
    assert np.abs(color.cpu().numpy() - f["color"]).max() <= 4e-3 * max(1.0, np.abs(f["rgb"]).max()), "a flip moves a pixel by <= alpha T c"
    assert (dg[f["n_contrib"] == 0] == 15.0).all()
    f["_flips_allowed"] = allow
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
        _grad_check(k, g[k].cpu().numpy(), b[k], f["_flips_allowed"], pure_min=PURE_REL_MIN_FRACTION_HUGE_SPLATS)


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
    f["_flips_allowed"] = _flip_allowance(int(f["n_contrib"].sum()))
    b = ro.backward(p, f, g_img, sc["means3D"], **okw)
    color, radii, depth, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], **gkw)
    g = R.rasterize_backward(rs, st, torch.tensor(g_img, device="cuda"), t["means3D"], **gkw)
    torch.cuda.synchronize()
    for k in ("means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"):
        if b[k] is None:
            assert g[k] is None, k
            continue
        _grad_check(k, g[k].cpu().numpy(), b[k], f["_flips_allowed"])


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
        _grad_check(k, g[k].cpu().numpy(), b[k], f["_flips_allowed"])


@pytest.mark.parametrize("N,W,H,deg,log_scale", [(1_500_000, 1920, 1080, 3, None), (60_000, 640, 400, 3, np.log(0.05)),
                                                 (20_000, 320, 200, 2, np.log(0.04)), (20_000, 320, 200, 1, np.log(0.04)),
                                                 (5_000, 160, 96, 0, np.log(0.1))])
def test_deferred_colours_equal_eager_colours_bit_for_bit(R, N, W, H, deg, log_scale):
    """SH -> RGB on first use in the render kernel (the default) against SH -> RGB for every visible Gaussian in the
    preprocess kernel (mvi_raster_color_mode(0)): image, depth, final_T and n_contrib identical bit for bit, every colour the
    deferred forward evaluated identical to the eager one, everything it left pending identical after resolve_colors();
    the evaluated set contains the gradient support (the backward replays only staged entries) and is a small part of the
    visible Gaussians at the headline size; gradients equal to summation order."""
    from multiview_inpaint_amd import _lib
    L = _lib.lib()
    cam = syn.make_camera(W, H, 50.0)
    kw = {} if log_scale is None else dict(log_scale_mean=log_scale)
    sc = syn.make_scene(N, cam, deg, seed=3, **kw)
    t = _to_dev(sc)
    bg = np.array([0.1, 0.3, 0.2], np.float32)
    rs = _settings(R, cam, bg, deg)
    g_img = torch.randn(3, H, W, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    okw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    out = {}
    prev = L.mvi_raster_color_mode(-1)
    try:
        for mode in (0, 1):
            L.mvi_raster_color_mode(mode)
            c, radii, d, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], prepare_backward=True, **okw)
            rgbd = st.tensor("rgbd", (N, 4), torch.float32)
            cl = st.tensor("clamped", (N,), torch.uint8)
            g = R.rasterize_backward(rs, st, g_img, t["means3D"], **okw)
            sup = st.tensor("grad_support", (N,), torch.uint8).bool()
            st.resolve_colors()
            out[mode] = dict(c=c, d=d, radii=radii, ft=st.tensor("final_T", (H, W), torch.float32),
                             nc=st.tensor("n_contrib", (H, W), torch.int32), rgbd=rgbd, cl=cl, g=g, sup=sup,
                             rgbd_all=st.tensor("rgbd", (N, 4), torch.float32), cl_all=st.tensor("clamped", (N,), torch.uint8))
    finally:
        L.mvi_raster_color_mode(prev)
    e, l = out[0], out[1]
    for k in ("c", "d", "radii", "ft", "nc"):
        assert torch.equal(e[k], l[k]), k
    vis = e["radii"] > 0
    assert not (e["rgbd"][vis][:, :3] < 0).any(), "the eager forward left a colour pending"
    done = vis & ~(l["rgbd"][:, :3] < 0).any(1)
    assert torch.equal(l["rgbd"][done], e["rgbd"][done]) and torch.equal(l["cl"][done], e["cl"][done])
    assert torch.equal(l["rgbd_all"][vis], e["rgbd"][vis]) and torch.equal(l["cl_all"][vis], e["cl"][vis])
    assert torch.equal(e["sup"], l["sup"]) and (done | ~l["sup"]).all(), "a Gaussian in the gradient support was never evaluated"
    if N >= 1_000_000:
        assert int(done.sum()) < 0.15 * int(vis.sum()), (int(done.sum()), int(vis.sum()))
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations"):
        assert _same_to_summation_order(l["g"][k].cpu().numpy(), e["g"][k].cpu().numpy()), k


def test_headline_config_1p5M_1080p_against_the_oracle(R, ro):
    """BASELINE.json configs[2] at FULL size (1.5 M Gaussians, 1920x1080, sh_degree 3 — the scene bench.py times) against
    oracle/raster_oracle.c: num_rendered, radii, tiles touched, point list, sort keys (tile ids) and ranges bit-exact, the
    per-Gaussian floats bit-exact, image / depth / n_contrib by the margin rule, all six gradient arrays elementwise 1e-4.
    The oracle's per-Gaussian and per-pixel loops run on the host's cores for this one test (forward results do not depend
    on the thread count; the backward sums each thread's band of tile rows in thread order, an fp32 reordering of the same
    kind as the GPU's float atomics, inside the elementwise bar's floor)."""
    import os
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cam = syn.make_camera(1920, 1080, 50.0)
    sc = syn.make_scene(1_500_000, cam, 3, seed=0)
    bg = np.zeros(3, np.float32)
    ro.set_threads(max(1, min(32, avail)))
    try:
        # marked pixels: 0.55 % here (a pixel of this scene walks ~5x the list entries of the 100k / 800x800 scene, each one
        # a chance to sit on a threshold); the differing ones among them are still counted against 2 + 1e-7 per pair
        f, p, rs, t, st = _check_forward(R, ro, cam, sc, bg, max_marginal=8e-3)
        assert f["num_rendered"] > 10_000_000 and (f["radii"] > 0).sum() > 1_000_000      # the case means what it says
        g_img = np.random.default_rng(0).normal(size=(3, 1080, 1920)).astype(np.float32)
        b = ro.backward(p, f, g_img, sc["means3D"], shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    finally:
        ro.set_threads(1)
    g = R.rasterize_backward(rs, st, torch.tensor(g_img, device="cuda"), t["means3D"], shs=t["shs"], scales=t["scales"],
                             rotations=t["rotations"])
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations"):
        _grad_check(k, g[k].cpu().numpy(), b[k], f["_flips_allowed"])


def _assert_binning_properties(st, radii, W, H):
    """Sorted (tile | depth) keys + payload permutation + stability + ranges: together they pin the binning output uniquely
    (a stable sort has one answer), so at sizes the oracle cannot reach they stand in for a bit-exact comparison."""
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


def test_full_size_properties_1p5M_1080p(R):
    """BASELINE.json configs[2] size (1.5M Gaussians, 1920x1080): size-independent properties."""
    cam = syn.make_camera(1920, 1080, 50.0)
    sc = syn.make_scene(1_500_000, cam, 3, seed=0)
    t = _to_dev(sc)
    W, H = 1920, 1080
    kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    bg0, bg1 = np.zeros(3, np.float32), np.array([0.25, 0.5, 1.0], np.float32)
    c0, radii, d0, st = R.rasterize_forward(_settings(R, cam, bg0, 3), t["means3D"], t["opacities"], **kw)
    _assert_binning_properties(st, radii, W, H)
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


@pytest.mark.parametrize("N,W,H", [(300_000, 640, 368), (2_300_000, 800, 448), (262_144, 320, 192)])
def test_binning_properties_at_mid_sizes(R, N, W, H):
    """Sizes between the oracle-checked scenes and the headline one (ragged last sort tiles in both levels, a power of two);
    twice each, on the same scratch buffers."""
    cam = syn.make_camera(W, H, 50.0)
    sc = syn.make_scene(N, cam, 0, seed=11)
    t = _to_dev(sc)
    kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    bg = np.zeros(3, np.float32)
    for _ in range(2):
        _, radii, _, st = R.rasterize_forward(_settings(R, cam, bg, 0), t["means3D"], t["opacities"], **kw)
        _assert_binning_properties(st, radii, W, H)



def _binning_outputs(R, cam, t, deg, version):
    from multiview_inpaint_amd import _lib
    L = _lib.lib()
    prev = L.mvi_raster_binning_version(version)
    try:
        W, H = cam["W"], cam["H"]
        _, radii, _, st = R.rasterize_forward(_settings(R, cam, np.zeros(3, np.float32), deg), t["means3D"], t["opacities"],
                                              shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        out = dict(D=st.D, radii=radii.clone(), plist=st.tensor("point_list", (st.D,), torch.int32),
                   tids=st.tensor("tile_ids_sorted", (st.D,), torch.int32),
                   ranges=st.tensor("ranges", (tiles, 2), torch.int32), id_bytes=st.views().tile_id_bytes)
        torch.cuda.synchronize()
        return out, st
    finally:
        L.mvi_raster_binning_version(prev)


@pytest.mark.parametrize("N,W,H,deg,log_scale", [
    (1_500_000, 1920, 1080, 3, None),          # the headline scene: 128-bin kernels, one image group per block
    (400_000, 3840, 2160, 0, None),            # 240 x 135 tiles: the 256-bin instantiation
    (200_000, 4096, 4096, 0, np.log(0.03)),    # 256 x 256 tiles: every bin index in use
    (9_000, 1920, 1080, 0, np.log(0.5)),       # huge footprints: a block's entries exceed the LDS image (several groups)
    (70_000, 800, 800, 1, np.log(0.08)),       # large footprints at the bring-up image size
    (5, 64, 48, 0, np.log(0.3)), (2049, 176, 112, 1, np.log(0.2))])
def test_binning_version_2_equals_version_1_bit_for_bit(R, N, W, H, deg, log_scale):
    """The rectangle-expanding partition (csrc/raster_binning2.hip: depth sort with in-kernel table sums, column pass, row
    pass + tile ranges) against the pair-emitting radix partition (csrc/raster_binning.hip, the round-1 / round-2 path that
    the oracle tests pinned): num_rendered, point list, tile ids and ranges identical, and version 2's output passes the
    size-independent binning properties on its own."""
    cam = syn.make_camera(W, H, 50.0)
    kw = {} if log_scale is None else dict(log_scale_mean=log_scale)
    sc = syn.make_scene(N, cam, deg, seed=5, **kw)
    t = _to_dev(sc)
    a, st2 = _binning_outputs(R, cam, t, deg, 2)
    b, _ = _binning_outputs(R, cam, t, deg, 1)
    assert a["D"] == b["D"] and a["D"] > 0
    assert torch.equal(a["radii"], b["radii"])
    assert torch.equal(a["plist"], b["plist"]), "point lists differ"
    assert torch.equal(a["tids"], b["tids"]), "tile ids differ"
    assert torch.equal(a["ranges"], b["ranges"]), "tile ranges differ"
    _assert_binning_properties(st2, a["radii"], W, H)


@pytest.mark.parametrize("seed,deg,pose,N,W,H", [(0, 3, True, 400, 100, 70), (3, 2, True, 5000, 320, 240)])
def test_forward_parity_small_with_binning_version_1(R, ro, seed, deg, pose, N, W, H):
    """The oracle comparison of test_forward_parity_small on the version-1 binning (still the path of grids above 256 x 256
    tiles)."""
    from multiview_inpaint_amd import _lib
    L = _lib.lib()
    prev = L.mvi_raster_binning_version(1)
    try:
        cam, sc, bg = small_scene(seed, N=N, W=W, H=H, deg=deg, pose=pose, log_scale=np.log(0.04))
        _check_forward(R, ro, cam, sc, bg)
    finally:
        L.mvi_raster_binning_version(prev)


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
    bad, worst = _viol(color.cpu().numpy(), f["color"], 1.0)
    fr = ro.margins(p, f)
    f["_flips_allowed"] = _flip_allowance(int(f["n_contrib"].sum()))
    assert not (bad.any(0) & ((fr & 1) == 0)).any() and bad.any(0).sum() <= f["_flips_allowed"], worst
    assert (g["shs"][:, 4:] == 0).all()                   # inactive coefficients get exactly zero gradient
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations"):
        _grad_check(k, g[k].cpu().numpy(), b[k], f["_flips_allowed"])


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
            assert _same_to_summation_order(gf[k].cpu().numpy(), gd[k].cpu().numpy()), k     # atomics: summation order only
        assert _same_to_summation_order(gf["sh_color_factor"].cpu().numpy(), gb["sh_color_factor"].cpu().numpy())
        # the degree-0 coefficient of the dense gradient is C0 * factor
        assert _same_to_summation_order((gb["sh_color_factor"] * 0.28209479177387814).cpu().numpy(), gb["shs"][:, 0].cpu().numpy(), 1e-6)
        dense.append(gb["shs"])
        factors.append(gb["sh_color_factor"])
        campos.append(rs.campos)
    both, cp = torch.stack(factors), torch.stack(campos)
    for n_views in (1, 2):
        out = R.sh_backward_views(t["means3D"], cp[:n_views], both[:n_views], M, deg)
        want = sum(dense[:n_views])
        assert out.shape == (P, M, 3) and _same_to_summation_order(out.cpu().numpy(), want.cpu().numpy(), 1e-5), n_views
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
    # the raw-parameter backward once more with the DENSE chain-rule kernel (the default is the sparse one): same gradients
    from multiview_inpaint_amd import _lib
    prev = _lib.lib().mvi_raster_backward_mode(1)
    try:
        _, _, _, g2 = run(True)
    finally:
        _lib.lib().mvi_raster_backward_mode(prev)
    for k in g1:
        if g1[k].numel():
            assert _same_to_summation_order(g2[k].cpu().numpy(), g1[k].cpu().numpy()), ("dense vs sparse", k)
    assert torch.equal(r0, r1) and int((r0 > 0).sum()) > P // 3
    assert torch.equal(c0, c1) and torch.equal(d0, d1)                 # forward has no atomics: identical
    for k in g0:
        assert g1[k] is not None and g1[k].shape == g0[k].shape, k
        if g0[k].numel() == 0:
            continue                                                   # degree 0: features_rest is [P,0,3]
        assert _same_to_summation_order(g1[k].cpu().numpy(), g0[k].cpu().numpy()), k


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


# ---- edge cases the per-quadrant cull and the clamps rely on --------------------------------------------------------------

def _fwd_bwd_vs_oracle(R, ro, cam, sc, bg, seed):
    f, p, rs, t, st = _check_forward(R, ro, cam, sc, bg)
    g_img = np.random.default_rng(seed).normal(size=(3, cam["H"], cam["W"])).astype(np.float32)
    kw = dict(shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    b = ro.backward(p, f, g_img, sc["means3D"], **kw)
    g = R.rasterize_backward(rs, st, torch.tensor(g_img, device="cuda"), t["means3D"], shs=t["shs"], scales=t["scales"],
                             rotations=t["rotations"])
    torch.cuda.synchronize()
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations"):
        _grad_check(k, g[k].cpu().numpy(), b[k], f["_flips_allowed"])
    return f


def test_edge_opacity_one_and_centres_on_quadrant_corners(R, ro):
    """Opacity exactly 1 (alpha clamps at 0.99), and Gaussians whose pixel centre sits exactly on the corner shared by
    four 8x8 quadrants (x = 8q - 0.5) or exactly on a pixel (x = 8q): the per-quadrant exact cull (quad_overlap) decides
    on the minimum of the quadratic form over the quadrant's box, and these centres put that minimum on the box edge."""
    W, H = 128, 96
    cam = syn.make_camera(W, H, 50.0)                        # camera at the origin looking down +z: ndc = X / (Z tanfov)
    sc = syn.make_scene(1400, cam, 1, seed=5, log_scale_mean=np.log(0.03), zmin=1.0, zmax=5.0)
    rng = np.random.default_rng(5)
    n = 700
    z = sc["means3D"][:n, 2]
    qx, qy = rng.integers(0, W // 8 + 1, n), rng.integers(0, H // 8 + 1, n)
    on_pixel = rng.random(n) < 0.5                           # half on the corner between pixels, half on the pixel 8q
    px = 8.0 * qx - np.where(on_pixel, 0.0, 0.5)
    py = 8.0 * qy - np.where(on_pixel, 0.0, 0.5)
    sc["means3D"][:n, 0] = (((2.0 * px + 1.0) / W - 1.0) * z * cam["tanfovx"]).astype(np.float32)
    sc["means3D"][:n, 1] = (((2.0 * py + 1.0) / H - 1.0) * z * cam["tanfovy"]).astype(np.float32)
    sc["opacities"][::3] = 1.0
    sc["scales"][:n:5] *= 6.0                                # some of them large enough to span many quadrants
    f = _fwd_bwd_vs_oracle(R, ro, cam, sc, np.array([0.1, 0.2, 0.3], np.float32), 5)
    on_corner = np.abs(f["xy"][:n] * 2 - np.round(f["xy"][:n] * 2)).max(1) < 1e-3
    assert on_corner.mean() > 0.9 and (f["radii"][:n] > 0).mean() > 0.8


def test_edge_transmittance_lands_on_one_half_exactly(R, ro):
    """The median-depth rule is `T > 0.5 and T (1 - alpha) < 0.5`, both strict (oracle/raster_oracle.c, orc_render_forward): an
    entry that leaves T at EXACTLY 0.5 is not a crossing and nothing behind it starts from above 0.5, so the pixel keeps the
    15.0 sentinel although its final T is below 0.5. Constructible: opacity 0.5 with the pixel centre on the mean (exp(0) = 1,
    alpha = 0.5, T = 1 -> 0.5 with no rounding), an opaque Gaussian behind it. The margin rule of the other tests marks such a
    pixel fragile and would excuse it; here it is asserted by name, next to a pixel that crosses properly."""
    W = H = 33                                                   # odd: ndc 0 is the centre of pixel 16
    cam = syn.make_camera(W, H, 50.0)
    z = lambda *s: np.zeros(s, np.float32)
    means = np.array([[0, 0, 2.0], [0, 0, 3.0], [0.35, 0.2, 2.5], [0.35, 0.2, 4.0]], np.float32)
    sc = dict(means3D=means, scales=np.float32([[0.08] * 3, [0.08] * 3, [0.2] * 3, [0.3] * 3]), rotations=np.tile(np.float32([1, 0, 0, 0]), (4, 1)),
              opacities=np.float32([[0.5], [0.9], [0.7], [0.9]]), shs=np.float32(np.random.default_rng(0).normal(0, 1, (4, 1, 3))),
              sh_degree=0)
    bg = z(3)
    p = oracle_params(ro, cam, sc, bg)
    f = ro.forward(p, sc["means3D"], sc["opacities"], shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    assert np.allclose(f["xy"][0], [16.0, 16.0], atol=1e-5) and f["final_T"][16, 16] < 0.5
    assert f["depth"][0, 16, 16] == 15.0, "the oracle itself must hold the strict rule"
    t = _to_dev(sc)
    color, radii, depth, st = R.rasterize_forward(_settings(R, cam, bg, 0), t["means3D"], t["opacities"], shs=t["shs"],
                                                  scales=t["scales"], rotations=t["rotations"])
    dg = depth.cpu().numpy()[0]
    assert st.tensor("final_T", (H, W), torch.float32).cpu().numpy()[16, 16] < 0.4      # the second entry took T well below 0.5
    assert dg[16, 16] == 15.0, "an entry that leaves T at exactly 0.5 must not be reported as the median"
    # the other pair (0.7 then 0.9) crosses at its first entry: depth 2.5 around its centre, identical to the oracle
    fr = ro.margins(p, f)
    assert np.array_equal(dg[fr == 0], f["depth"][0][fr == 0]) and (dg == 2.5).sum() > 4 and (f["depth"][0] == 2.5).sum() > 4


def test_edge_cov3D_precomp_huge_indefinite_and_degenerate(R, ro):
    """cov3D_precomp straight from the caller (compute_cov3D_python, gaussian_renderer/__init__.py:62-65) is not
    guaranteed to be a covariance: huge entries (radius >> image), an indefinite matrix (negative determinant of the 2-D
    projection: conic with negative entries, power > 0 everywhere) and a rank-deficient one (det == 0 -> skipped).
    Integers bit-exact, image and gradients within tolerance, nothing non-finite."""
    cam, sc, bg = small_scene(31, N=600, W=144, H=96, deg=0, log_scale=np.log(0.05))
    p = oracle_params(ro, cam, sc, bg)
    f0 = ro.forward(p, sc["means3D"], sc["opacities"], shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"], render=False)
    c6 = f0["cov3D"].copy()
    c6[0:20] *= 3e4                                         # huge footprints: clipped to the image, hundreds of tiles each
    c6[20:40, [0, 3, 5]] *= -1.0                            # negative diagonal: indefinite
    c6[40:60] = 0.0                                         # rank 0: only the +0.3 dilation is left
    c6[60:80, 1] = 10.0 * np.sqrt(np.abs(c6[60:80, 0] * c6[60:80, 3]))     # |xy| > sqrt(xx yy): indefinite, off-diagonal
    cp = np.abs(np.random.default_rng(31).normal(0.5, 0.3, (p.P, 3))).astype(np.float32)
    f = ro.forward(p, sc["means3D"], sc["opacities"], colors_precomp=cp, cov3D_precomp=c6)
    f["_flips_allowed"] = _flip_allowance(int(f["n_contrib"].sum()))
    rs = _settings(R, cam, bg, 0)
    t = _to_dev(sc)
    gkw = dict(colors_precomp=torch.tensor(cp, device="cuda"), cov3D_precomp=torch.tensor(c6, device="cuda"))
    color, radii, depth, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], **gkw)
    torch.cuda.synchronize()
    assert st.D == f["num_rendered"] and np.array_equal(radii.cpu().numpy(), f["radii"])
    assert np.array_equal(st.tensor("point_list", (st.D,), torch.int32).cpu().numpy().view(np.uint32), f["point_list"])
    assert (f["radii"][0:20] > 100).any()
    bad, worst = _viol(color.cpu().numpy(), f["color"], 1.0)
    fr = ro.margins(p, f)
    assert not (bad.any(0) & ((fr & 1) == 0)).any() and bad.any(0).sum() <= f["_flips_allowed"], worst
    assert torch.isfinite(color).all() and torch.isfinite(depth).all()
    g_img = np.random.default_rng(3).normal(size=(3, cam["H"], cam["W"])).astype(np.float32)
    b = ro.backward(p, f, g_img, sc["means3D"], colors_precomp=cp, cov3D_precomp=c6)
    g = R.rasterize_backward(rs, st, torch.tensor(g_img, device="cuda"), t["means3D"], **gkw)
    for k in ("means3D", "means2D", "opacities", "colors_precomp", "cov3D_precomp"):
        assert torch.isfinite(g[k]).all(), k
        got = g[k].cpu().numpy()
        # This scene is about binning and culling, not about gradient conditioning: twenty screen-filling Gaussians with
        # alpha up to 0.99 sit in every pixel's list, the replay divides T by (1 - alpha) per layer (x100 at the clamp), and
        # the doctored covariances have entries of 1e4 with determinants that cancel to a few ulp — fma-vs-separate rounding
        # moves such gradients by up to ~1e-4 of the array's scale for the ordinary Gaussians behind them and ~1e-2 of it
        # for the doctored ones. Bars here: 1e-3 / 1e-2 of the scale; the elementwise 1e-4 bar is held by every other test.
        ord_scale = np.abs(b[k][80:]).max() + 1e-30
        assert np.abs(got[80:] - b[k][80:]).max() <= 1e-3 * ord_scale, (k, float(np.abs(got[80:] - b[k][80:]).max() / ord_scale))
        scale = np.abs(b[k]).max() + 1e-30
        assert np.abs(got[:80] - b[k][:80]).max() <= 1e-2 * scale, (k, float(np.abs(got[:80] - b[k][:80]).max() / scale))


def test_edge_1080p_scale_image_not_a_multiple_of_the_tile(R, ro):
    """1912 x 1075: neither side a multiple of 16 (nor of the 8-pixel quadrants) at the headline image scale — the last
    tile column / row is partial, tile ids run to 120 x 68 - 1; forward and backward against the oracle."""
    cam = syn.make_camera(1912, 1075, 50.0)
    sc = syn.make_scene(120_000, cam, 1, seed=3)
    _fwd_bwd_vs_oracle(R, ro, cam, sc, np.array([0.2, 0.1, 0.4], np.float32), 3)


@pytest.mark.parametrize("N", [3000, 3001, 3002, 3003])
def test_ranged_backward_with_overlapped_exchange_equals_one_call(R, N):
    """(P % 4 in {0, 1, 2, 3}: the last range's sub-arrays must still start on 16-byte boundaries — the chain-rule kernel
    stores dL/drotations as float4.) rasterize_backward_ranged (render backward, then the chain rule in four Gaussian ranges through
    mvi_raster_backward_geom_range, each range's all-reduce started behind its kernel: dist.RangedGradExchange) on ONE rank
    over RCCL: the sums of one rank are the rank's own gradients, so everything must equal rasterize_backward to the order
    of the float atomics, dL/dSH included (rebuilt from the colour factor and the camera centre)."""
    import os
    import socket
    import torch.distributed as td
    from multiview_inpaint_amd import dist as md
    cam, sc, bg = small_scene(9, N=N, W=200, H=120, deg=3, pose=True, log_scale=np.log(0.05))
    t = _to_dev(sc)
    rs = _settings(R, cam, bg, 3)
    kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    color, radii, depth, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], **kw)
    g_img = torch.randn(3, cam["H"], cam["W"], device="cuda", generator=torch.Generator("cuda").manual_seed(4))
    one = R.rasterize_backward(rs, st, g_img, t["means3D"], **kw)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("nccl", rank=0, world_size=1)
    try:
        P, M = t["means3D"].shape[0], t["shs"].shape[1]
        ex = md.RangedGradExchange(P, M, 3, "cuda", n_ranges=4)
        assert len(ex.ranges) == 4 and all(a % 64 == 0 for a, _ in ex.ranges)
        assert all(v.data_ptr() % 16 == 0 for r in range(4) for v in ex.range_views(r).values())
        two = R.rasterize_backward_ranged(rs, st, g_img, t["means3D"], t["shs"], t["scales"], t["rotations"], ex)
        torch.cuda.synchronize()
    finally:
        td.destroy_process_group()
    for k in ("means3D", "means2D", "opacities", "scales", "rotations", "shs"):
        assert two[k].shape == one[k].shape, k
        assert _same_to_summation_order(two[k].cpu().numpy(), one[k].cpu().numpy()), k


def test_visibility_compacted_exchange_on_one_rccl_rank_equals_the_plain_backward(R):
    """dist.CompactedGradExchange on ONE rank over RCCL with the compaction forced (THRESHOLD = 1), keyed on the gradient
    support the render backward recorded: gather of the support's rows into capacity-sized buffers, all-reduce / all-gather of
    those, rebuild of dL/dSH for the rows, scatter back — the sums of one rank are the rank's own gradients, so everything
    equals rasterize_backward to fp32 rounding, and every row outside the support is exactly zero. Three steps over TWO
    different views on one exchange object: step 1 has no history (capacity = P, one page), step 2 (the other view) runs with
    the capacity step 1 left (1.25 x its union) and must clear the dL/dSH rows step 1 wrote, step 3 is forced to a capacity
    far below its union and has to take several pages."""
    import os
    import socket
    import torch.distributed as td
    from multiview_inpaint_amd import dist as md
    cam, sc, bg = small_scene(19, N=5003, W=200, H=120, deg=3, pose=True, log_scale=np.log(0.05))
    rng = np.random.default_rng(19)                                                    # small_scene's own pose, moved sideways
    R0, T0 = syn.random_rotation(rng), rng.normal(size=3)
    cam2 = syn.make_camera(200, 120, 50.0, R0, T0 + np.array([0.6, -0.2, 0.3]))
    t = _to_dev(sc)
    kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    g_img = torch.randn(3, cam["H"], cam["W"], device="cuda", generator=torch.Generator("cuda").manual_seed(5))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("nccl", rank=0, world_size=1)
    try:
        P, M = t["means3D"].shape[0], t["shs"].shape[1]
        ex = md.CompactedGradExchange(P, M, 3, "cuda")
        ex.THRESHOLD, ex.MIN_CAPACITY, ex.ROUND = 1.0, 64, 64
        fracs, prev_n = [], None
        for step, (c, force_cap) in enumerate(((cam, None), (cam2, None), (cam, 100))):
            rs = _settings(R, c, bg, 3)
            color, radii, depth, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], **kw)
            one = R.rasterize_backward(rs, st, g_img, t["means3D"], **kw)
            R.rasterize_backward(rs, st, g_img, t["means3D"], out=ex.views, sh_grad="factor", **kw)
            support = st.tensor("grad_support", (P,), torch.uint8)
            if force_cap is not None:
                ex._next_cap = force_cap
            got = ex.exchange_support(t["means3D"], rs.campos, support)
            torch.cuda.synchronize()
            assert ex.last_compacted and 0.0 < ex.last_union_fraction <= float((radii > 0).float().mean())
            n = int(support.sum())
            assert abs(ex.last_union_fraction * P - n) < 0.5
            if step == 0:
                assert ex.last_pages == 1 and ex.last_capacity == P
            elif step == 1:
                want = min(P, max(64, -(-int(1.25 * prev_n) // 64) * 64))
                assert ex.last_capacity == want and ex.last_pages == -(-n // want), (ex.last_pages, ex.last_capacity, want, n)
            else:
                assert ex.last_pages == -(-n // 100) and ex.last_pages > 3, (ex.last_pages, n)
            fracs.append(ex.last_union_fraction)
            prev_n = n
            vis = support.bool().cpu().numpy()
            for k in ("means3D", "opacities", "scales", "rotations", "shs"):
                a, b = got[k].cpu().numpy(), one[k].cpu().numpy()
                assert a.shape == b.shape, k
                assert _same_to_summation_order(a, b), (step, k)
                assert (a[~vis] == 0).all(), (step, k)
        assert fracs[0] != fracs[1], "the two views were meant to have different supports"
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("N,W,H,deg,mode,log_scale", [(60_000, 640, 368, 3, "sh", None), (3001, 200, 120, 1, "sh", np.log(0.05)),
                                                      (5000, 320, 240, 2, "precomp", np.log(0.04)), (400_000, 800, 448, 0, "sh", None)])
def test_sparse_backward_equals_dense_backward_and_the_support_is_a_superset(R, N, W, H, deg, mode, log_scale):
    """mvi_raster_backward's default form (outputs zeroed inside the render backward, chain rule only for the Gaussians whose
    accumulation row was touched: csrc/raster_preprocess.hip preprocess_backward_sparse_kernel) against the dense kernel:
    every gradient array equal to the order of the float atomics, every row outside the support exactly zero in both, and
    the support flags cover every non-zero row. (The oracle comparisons of this file run the sparse form: it is the default.)"""
    from multiview_inpaint_amd import _lib
    L = _lib.lib()
    cam = syn.make_camera(W, H, 50.0)
    kw0 = {} if log_scale is None else dict(log_scale_mean=log_scale)
    sc = syn.make_scene(N, cam, deg, seed=21, **kw0)
    t = _to_dev(sc)
    rs = _settings(R, cam, np.array([0.2, 0.1, 0.3], np.float32), deg)
    if mode == "precomp":
        cols = torch.rand(N, 3, device="cuda")
        fkw = dict(colors_precomp=cols, scales=t["scales"], rotations=t["rotations"])
        names = ("means3D", "means2D", "opacities", "colors_precomp", "scales", "rotations")
    else:
        fkw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        names = ("means3D", "means2D", "opacities", "shs", "scales", "rotations")
    color, radii, depth, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], **fkw)
    g_img = torch.randn(3, H, W, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    prev = L.mvi_raster_backward_mode(0)
    try:
        sparse = R.rasterize_backward(rs, st, g_img, t["means3D"], **fkw)
        support = st.tensor("grad_support", (N,), torch.uint8).bool()
        L.mvi_raster_backward_mode(1)
        dense = R.rasterize_backward(rs, st, g_img, t["means3D"], **fkw)
    finally:
        L.mvi_raster_backward_mode(prev)
    torch.cuda.synchronize()
    assert 0 < int(support.sum()) <= int((radii > 0).sum())
    for k in names:
        a, b = sparse[k], dense[k]
        assert a.shape == b.shape, k
        assert _same_to_summation_order(a.cpu().numpy(), b.cpu().numpy()), k
        flat_a, flat_b = a.reshape(N, -1), b.reshape(N, -1)
        assert (flat_a[~support] == 0).all() and (flat_b[~support] == 0).all(), k
    print(f"gradient support: {float(support.float().mean()):.4f} of the Gaussians (visible: {float((radii > 0).float().mean()):.3f})")


def test_sparse_backward_with_sh_rows_at_4_byte_alignment(R):
    """The chain rule's SH rows at dword alignment only — gradients written into dist.GradBucket's views (dL/dSH starts 12 P bytes
    into the flat buffer: P = 20001 leaves it at 12 mod 16) and the SH input itself one float into a larger buffer: the 16-byte
    accesses at dword alignment against the same call with 16-byte aligned arrays."""
    from multiview_inpaint_amd import dist as mdist
    N, W, H, deg = 20001, 320, 200, 3
    cam = syn.make_camera(W, H, 50.0)
    sc = syn.make_scene(N, cam, deg, seed=23, log_scale_mean=np.log(0.03))
    t = _to_dev(sc)
    rs = _settings(R, cam, np.array([0.2, 0.1, 0.3], np.float32), deg)
    big = torch.zeros(N * 48 + 1, device="cuda")
    shs_off = big[1:].view(N, 16, 3)
    shs_off.copy_(t["shs"])
    bucket = mdist.GradBucket(N, 16, torch.device("cuda"))
    assert shs_off.data_ptr() % 16 == 4 and bucket.views["shs"].data_ptr() % 16 == 12
    g_img = torch.randn(3, H, W, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    res = []
    for shs, out in ((t["shs"], None), (shs_off, bucket.views)):
        kw = dict(shs=shs, scales=t["scales"], rotations=t["rotations"])
        color, radii, depth, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], **kw)
        res.append(R.rasterize_backward(rs, st, g_img, t["means3D"], out=out, **kw))
    assert int((res[0]["shs"].abs().sum(dim=(1, 2)) > 0).sum()) > 500
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations"):
        assert _same_to_summation_order(res[1][k].cpu().numpy(), res[0][k].cpu().numpy()), k
    assert res[1]["shs"].data_ptr() == bucket.views["shs"].data_ptr()


@pytest.mark.parametrize("N,W,H,log_scale,squeeze", [
    (1, 16, 16, np.log(0.3), None), (5, 15, 33, np.log(0.3), None), (70, 4096, 16, np.log(0.05), None),
    (3000, 16, 4096, np.log(0.05), None), (3000, 33, 4090, np.log(0.02), None),
    (60_000, 1920, 1080, None, "column"),      # every Gaussian in ONE tile column: one bin holds every entry of pass 1
    (60_000, 1920, 1080, None, "row"),         # ... in one tile row: one bin holds every entry of pass 2
    (40_000, 640, 368, np.log(0.004), "tile"), # ... in one tile: a single 40 k-entry list
    (2048 * 3 + 1, 320, 192, np.log(0.3), None)])   # whole partition blocks + one Gaussian, footprints of hundreds of tiles
def test_binning_version_2_edge_shapes_and_skew_equal_version_1(R, N, W, H, log_scale, squeeze):
    """Shapes the partition kernels' geometry assumptions could break on — one-tile images, one tile column / row (gx or gy =
    1 or 256), entry counts far above the LDS image (several groups), and maximal skew (all entries in one bin: the bin's few
    owner threads walk everything) — version 2 against version 1, bit for bit, plus the binning properties."""
    cam = syn.make_camera(W, H, 50.0)
    kw = {} if log_scale is None else dict(log_scale_mean=log_scale)
    sc = syn.make_scene(N, cam, 0, seed=33, **kw)
    if squeeze:
        m = sc["means3D"].copy()
        if squeeze in ("column", "tile"):
            m[:, 0] *= 0.004                      # x within a few pixels of the image centre
        if squeeze in ("row", "tile"):
            m[:, 1] *= 0.004
        sc["means3D"] = m
    t = _to_dev(sc)
    a, st2 = _binning_outputs(R, cam, t, 0, 2)
    b, _ = _binning_outputs(R, cam, t, 0, 1)
    assert a["D"] == b["D"]
    assert torch.equal(a["radii"], b["radii"])
    assert torch.equal(a["plist"], b["plist"]), "point lists differ"
    assert torch.equal(a["tids"], b["tids"]), "tile ids differ"
    assert torch.equal(a["ranges"], b["ranges"]), "tile ranges differ"
    if a["D"]:
        _assert_binning_properties(st2, a["radii"], W, H)
