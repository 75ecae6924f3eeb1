"""dropin.patch_gs_simp's render (the model's STORED parameters into GaussianRasterizer.forward_raw) against a stand-in of the
reference's render() call shape (gaussian_renderer/__init__.py:18-104: PyTorch activations of gaussian_model.py:95-115, then the
rasterizer's standard entry): same dict, same integers, images / depth / gradients to the tolerance of the device activations, and
every call the patched form cannot serve lands in the reference's own function."""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "multiview_inpaint_amd", "dropin")
if DROPIN not in sys.path:
    sys.path.insert(0, DROPIN)


class Model:
    """The attributes and getters render() reads from scene.gaussian_model.GaussianModel (gaussian_model.py:24-115)."""

    def __init__(self, sc, max_deg, active_deg):
        dev = "cuda"
        t = {k: torch.tensor(v, device=dev) for k, v in sc.items() if k != "sh_degree"}
        self.max_sh_degree, self.active_sh_degree = max_deg, active_deg
        self.scaling_activation, self.opacity_activation = torch.exp, torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize
        P = t["means3D"].shape[0]
        g = torch.Generator(dev).manual_seed(3)
        self._xyz = torch.nn.Parameter(t["means3D"].clone())
        self._features_dc = torch.nn.Parameter(t["shs"][:, :1].contiguous())
        self._features_rest = torch.nn.Parameter(t["shs"][:, 1:].contiguous())
        self._opacity = torch.nn.Parameter(torch.logit(t["opacities"].clamp(1e-4, 1 - 1e-4)))
        self._scaling = torch.nn.Parameter(torch.log(t["scales"]))
        self._rotation = torch.nn.Parameter(t["rotations"] * (0.5 + torch.rand(P, 1, device=dev, generator=g)))

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: s.scaling_activation(s._scaling))
    get_rotation = property(lambda s: s.rotation_activation(s._rotation))
    get_opacity = property(lambda s: s.opacity_activation(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))

    def params(self):
        return dict(xyz=self._xyz, dc=self._features_dc, rest=self._features_rest, o=self._opacity, s=self._scaling, q=self._rotation)


class Camera:
    def __init__(self, cam):
        self.image_height, self.image_width = cam["H"], cam["W"]
        self.FoVx, self.FoVy = 2 * math.atan(cam["tanfovx"]), 2 * math.atan(cam["tanfovy"])
        self.world_view_transform = torch.tensor(cam["viewmatrix"], device="cuda")
        self.full_proj_transform = torch.tensor(cam["projmatrix"], device="cuda")
        self.camera_center = torch.tensor(cam["campos"], device="cuda")


class Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False


CALLS = []


def standin_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """The reference's call shape: activations in PyTorch, SH rows concatenated, the rasterizer's standard entry."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    CALLS.append("reference")
    pts = torch.zeros_like(pc.get_xyz, requires_grad=True) + 0
    if pts.requires_grad:
        pts.retain_grad()
    rs = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center,
        prefiltered=False)
    kw = dict(shs=pc.get_features, colors_precomp=None) if override_color is None else dict(shs=None, colors_precomp=override_color)
    img, radii, depth = GaussianRasterizer(raster_settings=rs)(means3D=pc.get_xyz, means2D=pts, opacities=pc.get_opacity,
                                                               scales=pc.get_scaling, rotations=pc.get_rotation, cov3D_precomp=None, **kw)
    return {"render": img, "depth": depth, "viewspace_points": pts, "visibility_filter": radii > 0, "radii": radii}


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("max_deg,active_deg,scale_mod", [(3, 3, 1.0), (3, 1, 1.0), (2, 2, 0.7), (0, 0, 1.0)])
def test_patched_render_equals_the_reference_call_shape(max_deg, active_deg, scale_mod):
    from multiview_inpaint_amd.dropin import patch_gs_simp
    from raster_helpers import small_scene
    cam, sc, bg = small_scene(33, N=4000, W=208, H=120, deg=max_deg, pose=True, log_scale=np.log(0.05))
    camera, pipe = Camera(cam), Pipe()
    bg_t = torch.tensor(bg, device="cuda")
    patched = patch_gs_simp._make_render(standin_render, Model)
    g_img = torch.randn(3, cam["H"], cam["W"], device="cuda", generator=torch.Generator("cuda").manual_seed(9))

    def run(fn):
        pc = Model(sc, max_deg, active_deg)
        pkg = fn(camera, pc, pipe, bg_t, scale_mod)
        (pkg["render"] * g_img).sum().backward()
        return pkg, {k: v.grad for k, v in pc.params().items()}

    CALLS.clear()
    want, gw = run(standin_render)
    assert CALLS == ["reference"]
    got, gg = run(patched)
    assert CALLS == ["reference"], "the patched render went through the reference's function"
    assert list(got.keys()) == list(want.keys())
    assert torch.equal(got["radii"], want["radii"]) and torch.equal(got["visibility_filter"], want["visibility_filter"])
    assert int(got["visibility_filter"].sum()) > 1000
    # torch.exp / sigmoid / normalize against the device expressions of the preprocess kernel: a few ulp on the inputs of the projection
    assert rel(got["render"], want["render"]) < 1e-4
    finite = torch.isfinite(want["depth"]) & (want["depth"] < 14.9)
    assert rel(got["depth"][finite], want["depth"][finite]) < 1e-4
    assert got["viewspace_points"].grad is not None and got["viewspace_points"].shape == want["viewspace_points"].shape
    assert rel(got["viewspace_points"].grad, want["viewspace_points"].grad) < 1e-3
    for k in gw:
        if gw[k] is None or gw[k].numel() == 0:                          # degree 0: features_rest is [P, 0, 3]
            continue
        assert gg[k] is not None and rel(gg[k], gw[k]) < 1e-3, k


def test_calls_the_patched_render_cannot_serve_go_to_the_reference():
    from multiview_inpaint_amd.dropin import patch_gs_simp
    from raster_helpers import small_scene
    cam, sc, bg = small_scene(34, N=500, W=64, H=48, deg=1, pose=True, log_scale=np.log(0.05))
    camera, bg_t = Camera(cam), torch.tensor(bg, device="cuda")
    patched = patch_gs_simp._make_render(standin_render, Model)
    pc = Model(sc, 1, 1)
    CALLS.clear()
    with torch.no_grad():
        patched(camera, pc, Pipe(), bg_t)
        assert CALLS == []
        patched(camera, pc, Pipe(), bg_t, 1.0, torch.rand(500, 3, device="cuda"))            # override_color
        assert CALLS == ["reference"]
        p2 = Pipe()
        p2.convert_SHs_python = True
        patched(camera, pc, p2, bg_t)
        assert CALLS == ["reference"] * 2

        class Sub(Model):                                                # a getter of its own: not what forward_raw computes
            get_opacity = property(lambda s: torch.sigmoid(s._opacity) * 0.5)
        patched(camera, Sub(sc, 1, 1), Pipe(), bg_t)
        assert CALLS == ["reference"] * 3
        pc3 = Model(sc, 1, 1)
        pc3.scaling_activation = torch.nn.functional.softplus            # another activation
        patched(camera, pc3, Pipe(), bg_t)
        assert CALLS == ["reference"] * 4

        class Plain(Model):                                              # subclasses with the stock getters stay on the fused path
            pass
        patched(camera, Plain(sc, 1, 1), Pipe(), bg_t)
        assert CALLS == ["reference"] * 4


def test_patched_densification_stats_equal_the_masked_form():
    """gaussian_model.py:482-484 restated with boolean-mask indexing against the patched mask-free form, three iterations with
    different filters; an index-tensor filter goes to the reference's method."""
    from multiview_inpaint_amd.dropin import patch_gs_simp
    calls = []

    class M:
        def __init__(self, P):
            self.xyz_gradient_accum = torch.zeros(P, 1, device="cuda")
            self.denom = torch.zeros(P, 1, device="cuda")

        def add_densification_stats(self, viewspace_point_tensor, update_filter):
            calls.append(1)
            self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
            self.denom[update_filter] += 1

    P = 100_003
    patched = patch_gs_simp._make_stats(M.add_densification_stats)
    a, b = M(P), M(P)
    g = torch.Generator("cuda").manual_seed(4)
    for it in range(3):
        pts = torch.zeros(P, 3, device="cuda", requires_grad=True)
        pts.grad = torch.randn(P, 3, device="cuda", generator=g) * 10.0 ** (it - 3)
        filt = torch.rand(P, device="cuda", generator=g) < (0.1, 0.8, 0.0)[it]
        M.add_densification_stats(a, pts, filt)
        n = len(calls)
        patched(b, pts, filt)
        assert len(calls) == n
        assert torch.equal(a.denom, b.denom)
        assert torch.equal(a.xyz_gradient_accum, b.xyz_gradient_accum)
    assert float(b.denom.sum()) > 0.85 * P
    idx = torch.arange(0, P, 7, device="cuda")
    n = len(calls)
    patched(b, pts, idx)
    assert len(calls) == n + 1


def test_short_training_run_matches_the_pytorch_formulation():
    """The pieces composed the way train.py composes them (train.py:86-128): 40 iterations of render -> (1 - l) * l1_loss + l * (1 -
    ssim) -> backward -> densification statistics -> Adam on a small scene, once with everything the import hook installs (render on
    the stored parameters, the paired loss functions, the mask-free statistics, FusedAdam) and once in the reference's own PyTorch
    formulation (activations / cat in PyTorch, loss_utils-style SSIM with depthwise convolutions, boolean-mask statistics,
    torch.optim.Adam) — both on the HIP rasterizer. The loss curves and the final parameters agree, and the loss goes down."""
    from multiview_inpaint_amd import train_ops as T
    from multiview_inpaint_amd.bench_train import torch_loss
    from multiview_inpaint_amd.dropin import patch_gs_simp
    from raster_helpers import small_scene
    cam, sc, bg = small_scene(35, N=5000, W=256, H=160, deg=2, pose=True, log_scale=np.log(0.06))
    camera, pipe, bg_t = Camera(cam), Pipe(), torch.tensor(bg, device="cuda")
    with torch.no_grad():                                    # the target: the same scene with other colours and opacities
        tgt = Model(sc, 2, 2)
        g = torch.Generator("cuda").manual_seed(8)
        tgt._features_dc.add_(0.5 * torch.randn(tgt._features_dc.shape, device="cuda", generator=g))
        tgt._opacity.add_(torch.randn(tgt._opacity.shape, device="cuda", generator=g))
        gt_image = standin_render(camera, tgt, pipe, bg_t)["render"].clone()
    lrs = dict(xyz=1.6e-4, dc=2.5e-3, rest=2.5e-3 / 20, o=0.05, s=5e-3, q=1e-3)         # arguments/__init__.py defaults

    class RefStats:
        def add_densification_stats(self, viewspace_point_tensor, update_filter):      # gaussian_model.py:482-484
            self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
            self.denom[update_filter] += 1

    def run(fused):
        pc = Model(sc, 2, 2)
        pc.xyz_gradient_accum, pc.denom = torch.zeros(5000, 1, device="cuda"), torch.zeros(5000, 1, device="cuda")
        groups = [{"params": [p], "lr": lrs[k], "name": k} for k, p in pc.params().items()]
        opt = (T.FusedAdam if fused else torch.optim.Adam)(groups, lr=0.0, eps=1e-15)
        render = patch_gs_simp._make_render(standin_render, Model) if fused else standin_render
        stats = patch_gs_simp._make_stats(RefStats.add_densification_stats) if fused else RefStats.add_densification_stats
        losses = []
        for it in range(40):
            pkg = render(camera, pc, pipe, bg_t)
            image = pkg["render"]
            if fused:
                loss = (1.0 - 0.2) * T.l1_loss(image, gt_image) + 0.2 * (1.0 - T.ssim(image, gt_image))
            else:
                loss = torch_loss(image, gt_image, 0.2)
            loss.backward()
            with torch.no_grad():
                losses.append(loss.item())
                stats(pc, pkg["viewspace_points"], pkg["visibility_filter"])
                opt.step()
                opt.zero_grad(set_to_none=True)
        return losses, {k: v.detach().clone() for k, v in pc.params().items()}, pc.xyz_gradient_accum, pc.denom

    CALLS.clear()
    la, pa, acc_a, den_a = run(True)
    assert CALLS == []                                       # the fused loop never entered the reference-shaped render
    lb, pb, acc_b, den_b = run(False)
    assert la[-1] < 0.8 * la[0], (la[0], la[-1])
    worst = max(abs(a - b) / b for a, b in zip(la, lb))
    print(f"loss {la[0]:.5f} -> {la[-1]:.5f}; worst relative difference of the two loss curves {worst:.2e}")
    assert worst < 3e-4, worst                               # (measured 3e-5: ten times that)
    # parameters: in rms — with eps = 1e-15 Adam moves a parameter whose gradient is noise by +-lr per step whatever the gradient's size, so a
    # handful of (occluded) Gaussians legitimately end a few lr apart between two formulations that differ in the last bits. What a
    # wrong update rule would do instead is move MANY elements apart: at most one element in a thousand may end more than 4 lr from
    # its twin (40 steps could carry it 80 lr away), and the rms distance stays below 5e-3 of the parameters' own rms
    for k in pa:
        if pa[k].numel():
            d, ref = (pa[k] - pb[k]).double(), pb[k].double()
            assert float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()) < 5e-3, k
            far = float((d.abs() > 4 * lrs[k]).double().mean())
            print(f"{k}: fraction of elements more than 4 lr apart {far:.2e}, max distance {float(d.abs().max()) / lrs[k]:.1f} lr")
            assert far < 1e-3, (k, far)
    assert torch.equal(den_a, den_b)
    assert rel(acc_a, acc_b) < 5e-3


def test_patched_prune_points_equals_the_boolean_index_form():
    """gaussian_model.py:335-365 restated (boolean indexing of parameters, Adam moments and statistics, optimizer.state re-keyed)
    against the patched prune_points (one mask scan + one gather launch): every tensor identical, the optimizer steps on; a mask
    that is an index tensor goes to the reference's method."""
    from multiview_inpaint_amd import train_ops as T
    from multiview_inpaint_amd.dropin import patch_gs_simp
    names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
    attrs = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
    shapes = dict(xyz=(3,), f_dc=(1, 3), f_rest=(15, 3), opacity=(1,), scaling=(3,), rotation=(4,))
    calls = []

    class M:
        def __init__(self, P, opt_cls):
            g = torch.Generator("cuda").manual_seed(2)
            for a, n in zip(attrs, names):
                setattr(self, a, torch.nn.Parameter(torch.randn(P, *shapes[n], device="cuda", generator=g)))
            self.optimizer = opt_cls([{"params": [getattr(self, a)], "lr": 1e-2, "name": n} for a, n in zip(attrs, names)], lr=0.0, eps=1e-15)
            for a in attrs:
                getattr(self, a).grad = torch.randn(getattr(self, a).shape, device="cuda", generator=g)
            self.optimizer.step()                                    # non-trivial moments
            self.xyz_gradient_accum = torch.rand(P, 1, device="cuda", generator=g)
            self.denom = torch.rand(P, 1, device="cuda", generator=g)
            self.max_radii2D = torch.rand(P, device="cuda", generator=g)

        def _prune_optimizer(self, mask):                            # gaussian_model.py:335-349
            out = {}
            for group in self.optimizer.param_groups:
                st = self.optimizer.state.get(group["params"][0], None)
                st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][mask], st["exp_avg_sq"][mask]
                del self.optimizer.state[group["params"][0]]
                group["params"][0] = torch.nn.Parameter(group["params"][0][mask].requires_grad_(True))
                self.optimizer.state[group["params"][0]] = st
                out[group["name"]] = group["params"][0]
            return out

        def prune_points(self, mask):                                # gaussian_model.py:351-365
            calls.append(1)
            valid = ~mask
            t = self._prune_optimizer(valid)
            for a, n in zip(attrs, names):
                setattr(self, a, t[n])
            self.xyz_gradient_accum, self.denom, self.max_radii2D = self.xyz_gradient_accum[valid], self.denom[valid], self.max_radii2D[valid]

    P = 50_001
    patched = patch_gs_simp._make_prune_points(M.prune_points)
    a, b = M(P, T.FusedAdam), M(P, T.FusedAdam)
    mask = torch.rand(P, device="cuda", generator=torch.Generator("cuda").manual_seed(3)) < 0.37
    M.prune_points(a, mask)
    n = len(calls)
    patched(b, mask)
    assert len(calls) == n, "the patched method went through the reference's"
    kept = int((~mask).sum())
    for attr in attrs + ("xyz_gradient_accum", "denom", "max_radii2D"):
        x, y = getattr(a, attr), getattr(b, attr)
        assert x.shape[0] == kept and torch.equal(x, y), attr
    for ga, gb in zip(a.optimizer.param_groups, b.optimizer.param_groups):
        pa, pb = ga["params"][0], gb["params"][0]
        assert isinstance(pb, torch.nn.Parameter) and pb.requires_grad and gb["name"] == ga["name"]
        sa, sb = a.optimizer.state[pa], b.optimizer.state[pb]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]) and float(sa["step"]) == float(sb["step"])
        assert pb is getattr(b, attrs[names.index(gb["name"])])
    for m in (a, b):                                                 # and training goes on from the pruned state
        for attr in attrs:
            getattr(m, attr).grad = torch.ones_like(getattr(m, attr))
        m.optimizer.step()
    assert torch.equal(a._xyz, b._xyz) and torch.equal(a._features_rest, b._features_rest)
    idx = torch.arange(0, kept, 5, device="cuda")
    n = len(calls)
    try:
        patched(b, idx)                                              # not a boolean mask: the reference's method (which rejects ~idx its own way)
    except Exception:
        pass
    assert len(calls) == n + 1


def test_patched_densification_equals_the_reference_recipe():
    """The densify half behind the hook: densify_and_prune of the stand-in model (tests/gs_standin.py: the
    reference's recipe, gaussian_model.py:384-480 — clone, split with torch.normal, cat_tensors_to_optimizer through
    densification_postfix, prune) run twice from identical states and identical generator seeds, once as written and once with
    dropin.patch_gs_simp's cat_tensors_to_optimizer + prune_points on the class: every parameter, both Adam moments and the
    statistics identical, optimizer.state re-keyed to the new parameters, and training steps on identically. A dict the hook does
    not serve (a missing group) lands in the reference's method."""
    from multiview_inpaint_amd import train_ops as T
    from multiview_inpaint_amd.dropin import patch_gs_simp
    from raster_helpers import small_scene
    import gs_standin as GS
    cam, sc, bg = small_scene(41, N=30_000, W=64, H=64, deg=2, pose=False, log_scale=np.log(0.05))
    calls = []

    class Ref(GS.StandinGaussianModel):
        def cat_tensors_to_optimizer(self, d):
            calls.append("cat")
            return GS.StandinGaussianModel.cat_tensors_to_optimizer(self, d)

    class Hooked(Ref):
        cat_tensors_to_optimizer = patch_gs_simp._make_cat_tensors(Ref.cat_tensors_to_optimizer)
        prune_points = patch_gs_simp._make_prune_points(GS.StandinGaussianModel.prune_points)

    def prepared(cls):
        m = cls(sc, 2, optimizer_cls=T.FusedAdam)
        g = torch.Generator("cuda").manual_seed(4)
        for p in m.params().values():                                # non-trivial moments
            p.grad = torch.randn(p.shape, device="cuda", generator=g)
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
        P = m._xyz.shape[0]
        m.xyz_gradient_accum = torch.rand(P, 1, device="cuda", generator=g) * 6e-4      # about a third above the threshold
        m.denom = torch.ones(P, 1, device="cuda")
        m.max_radii2D = torch.rand(P, device="cuda", generator=g) * 30
        return m
    a, b = prepared(Ref), prepared(Hooked)
    for m in (a, b):
        torch.manual_seed(9)
        torch.cuda.manual_seed(9)
        m.densify_and_prune(0.0002, 0.005, 6.0, 20)
    assert calls.count("cat") == 2, calls                          # clone + split of the reference run only: the hook served its own
    P2 = a._xyz.shape[0]
    assert P2 != 30_000 and b._xyz.shape[0] == P2
    for attr in GS.StandinGaussianModel.ATTRS + ("xyz_gradient_accum", "denom", "max_radii2D"):
        assert torch.equal(getattr(a, attr), getattr(b, attr)), attr
    for ga, gb in zip(a.optimizer.param_groups, b.optimizer.param_groups):
        pa, pb = ga["params"][0], gb["params"][0]
        assert isinstance(pb, torch.nn.Parameter) and pb.requires_grad and pb is getattr(b, GS.StandinGaussianModel.ATTRS[GS.StandinGaussianModel.NAMES.index(gb["name"])])
        sa, sb = a.optimizer.state[pa], b.optimizer.state[pb]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]) and float(sa["step"]) == float(sb["step"])
    for m in (a, b):
        for p in m.params().values():
            p.grad = torch.ones_like(p)
        m.optimizer.step()
    assert torch.equal(a._xyz, b._xyz) and torch.equal(a._features_rest, b._features_rest) and torch.equal(a._opacity, b._opacity)
    n = len(calls)
    try:
        Hooked.cat_tensors_to_optimizer(b, {"xyz": b._xyz[:2].detach()})          # not the six groups: the reference's method (which raises its own way)
    except Exception:
        pass
    assert len(calls) == n + 1
