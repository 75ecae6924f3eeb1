"""One full 3DGS training iteration on the GPU (the region gs-simp/train.py times between iter_start and iter_end plus
the optimizer step): parameter activations -> rasterize -> L1 + DSSIM loss -> backward -> Adam, N = 1.5 M Gaussians,
1920x1080, sh_degree 3, synthetic scene. Two variants of everything AROUND the HIP rasterizer:
  hip_raw: raw parameters straight into the rasterizer (GaussianRasterizer.forward_raw), fused loss, fused Adam
  hip   : multiview_inpaint_amd.train_ops (fused activations, fused loss, fused Adam)
  patched: what an UNCHANGED gs-simp/train.py gets under `python -m multiview_inpaint_amd.dropin.patch_gs_simp train.py ...`:
          gaussian_renderer.render handing the model's stored parameters to forward_raw (round 4; before: the reference's own
          exp / normalize / sigmoid / cat), its loss expression (1 - l) * l1_loss(image, gt) + l * (1 - ssim(image, gt)) with the
          two functions swapped for train_ops' (two calls that share ONE node: one statistics and one gradient pass), FusedAdam in place of the torch.optim.Adam its
          training_setup creates
  torch : the reference's own PyTorch-ROCm formulation (exp / normalize / sigmoid / cat, loss_utils-style SSIM with
          five depthwise convs, torch.optim.Adam)
Usage (GPU box): python -m multiview_inpaint_amd.bench_train [--steps 20]; bench.py reports the same numbers as
"train_iteration"."""
import argparse
import json

import torch
import torch.nn.functional as F

from . import raster as R, synthetic as syn, train_ops as T


def torch_loss(img, gt, lam=0.2):
    w1 = torch.tensor([pow(2.718281828459045, -(x - 5) ** 2 / 4.5) for x in range(11)], device=img.device)
    w1 = w1 / w1.sum()
    win = (w1[:, None] @ w1[None, :]).expand(3, 1, 11, 11).contiguous()
    conv = lambda t: F.conv2d(t[None], win, padding=5, groups=3)[0]
    mu1, mu2 = conv(img), conv(gt)
    s1, s2, s12 = conv(img * img) - mu1 * mu1, conv(gt * gt) - mu2 * mu2, conv(img * gt) - mu1 * mu2
    smap = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    return 0.8 * (img - gt).abs().mean() + 0.2 * (1 - smap.mean())


def run(variant, steps, warmup, N=1_500_000, W=1920, H=1080, deg=3, extras=None):
    dev = torch.device("cuda")
    cam = syn.make_camera(W, H, 50.0)
    sc = syn.make_scene(N, cam, deg, seed=0)
    t = {k: torch.tensor(v, device=dev) for k, v in sc.items() if k != "sh_degree"}
    raw = dict(xyz=t["means3D"], f_dc=t["shs"][:, :1].contiguous(), f_rest=t["shs"][:, 1:].contiguous(),
               opacity=torch.logit(t["opacities"].clamp(1e-4, 1 - 1e-4)), scaling=torch.log(t["scales"]), rotation=t["rotations"])
    prm = {k: torch.nn.Parameter(v.clone()) for k, v in raw.items()}
    lrs = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=2.5e-3 / 20, opacity=0.05, scaling=5e-3, rotation=1e-3)
    groups = [{"params": [prm[k]], "lr": lrs[k], "name": k} for k in prm]
    opt = (T.FusedAdam if variant != "torch" else torch.optim.Adam)(groups, lr=0.0, eps=1e-15)
    rs = R.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
        scale_modifier=1.0, viewmatrix=torch.tensor(cam["viewmatrix"], device=dev),
        projmatrix=torch.tensor(cam["projmatrix"], device=dev), sh_degree=deg, campos=torch.tensor(cam["campos"], device=dev),
        prefiltered=False)
    rast = R.GaussianRasterizer(rs)
    gt = torch.rand(3, H, W, device=dev, generator=torch.Generator(dev).manual_seed(1))

    # extras: what gs-simp/train.py does per iteration besides the timed region's kernels while it densifies (train.py:96, :113-116):
    # the loss read back for the progress bar, the max-radii update (boolean-mask indexing, in the script itself) and
    # add_densification_stats — "masked" = the reference's method, "patched" = dropin.patch_gs_simp's mask-free form
    stats = None
    if extras:
        from .dropin import patch_gs_simp

        class _Stats:
            def __init__(self):
                self.xyz_gradient_accum = torch.zeros(N, 1, device=dev)
                self.denom = torch.zeros(N, 1, device=dev)
                self.max_radii2D = torch.zeros(N, device=dev)

            def add_densification_stats(self, viewspace_point_tensor, update_filter):      # gaussian_model.py:482-484
                self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
                self.denom[update_filter] += 1
        stats = _Stats()
        add_stats = patch_gs_simp._make_stats(_Stats.add_densification_stats) if extras == "patched" else _Stats.add_densification_stats

    def loop_extras(loss, means2D, radii):
        if stats is None:
            return
        loss.item()
        vis = radii > 0
        stats.max_radii2D[vis] = torch.max(stats.max_radii2D[vis], radii[vis].to(stats.max_radii2D.dtype))
        add_stats(stats, means2D, vis)

    def step():
        if variant in ("hip_raw", "patched"):
            if variant == "patched":                            # the patched render(): gaussian_renderer/__init__.py:27 as written
                means2D = torch.zeros_like(prm["xyz"], requires_grad=True) + 0
                means2D.retain_grad()
            else:
                means2D = torch.zeros_like(prm["xyz"], requires_grad=True)
            image, radii, depth = rast.forward_raw(prm["xyz"], means2D, prm["f_dc"], prm["f_rest"], prm["opacity"],
                                                   prm["scaling"], prm["rotation"])
            if variant == "patched":                            # gs-simp/train.py:91-92 with the patched names
                loss = (1.0 - 0.2) * T.l1_loss(image, gt) + 0.2 * (1.0 - T.ssim(image, gt))
            else:
                loss = T.fused_l1_dssim_loss(image, gt, 0.2)
            loss.backward()
            loop_extras(loss, means2D, radii)
            opt.step()
            opt.zero_grad(set_to_none=True)
            return loss
        if variant == "hip":
            scales, rots, opac, shs = T.activate_gaussians(prm["scaling"], prm["rotation"], prm["opacity"], prm["f_dc"], prm["f_rest"])
        else:
            scales, rots, opac = torch.exp(prm["scaling"]), F.normalize(prm["rotation"]), torch.sigmoid(prm["opacity"])
            shs = torch.cat((prm["f_dc"], prm["f_rest"]), dim=1)
        means2D = torch.zeros_like(prm["xyz"], requires_grad=True)
        image, radii, depth = rast(means3D=prm["xyz"], means2D=means2D, shs=shs, colors_precomp=None, opacities=opac,
                                   scales=scales, rotations=rots, cov3D_precomp=None)
        if variant == "hip":
            loss = T.fused_l1_dssim_loss(image, gt, 0.2)
        else:
            loss = torch_loss(image, gt)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        loss = step()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    out = dict(variant=variant, ms_per_iteration=round(ms, 3), iterations_per_s=round(1e3 / ms, 1), loss=round(loss.item(), 5))
    if extras:
        out["extras"] = extras
    return out


def run_both(steps=20, warmup=3):
    out = [run(v, steps, warmup) for v in ("hip_raw", "hip", "patched", "torch")]
    return {"workload": "full 3DGS training iteration (activations, rasterize, L1+DSSIM, backward, Adam), N=1.5M, 1920x1080, "
                        "sh_degree 3; the HIP rasterizer in both variants, the ops around it fused HIP vs PyTorch-ROCm ops",
            "variants": {"hip_raw": "raw parameters into the rasterizer (activations + SH concat inside the preprocess kernels), "
                                    "fused loss, fused Adam", "hip": "fused activation kernel + standard rasterizer entry, fused "
                                    "loss, fused Adam", "patched": "an unchanged train.py under dropin.patch_gs_simp: render() patched to hand the stored "
                         "parameters to the rasterizer, l1_loss + ssim swapped for train_ops' (two calls, ONE autograd node), FusedAdam", "torch": "PyTorch-ROCm ops for activations, loss and Adam"},
            "results": out, "speedup_of_the_surrounding_ops": round(out[3]["ms_per_iteration"] / out[0]["ms_per_iteration"], 2)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--loop-extras", action="store_true",
                    help="instead: the patched iteration WITH train.py's per-iteration extras while densifying (loss.item(), max-radii "
                         "update, add_densification_stats), the reference's masked statistics against the patched mask-free form")
    ap.add_argument("--variants", default="", help="comma-separated subset of hip_raw,hip,patched,torch (for profiling one of them)")
    a = ap.parse_args()
    if a.variants:
        print(json.dumps({"results": [run(v, a.steps, a.warmup) for v in a.variants.split(",")]}))
    elif a.loop_extras:
        print(json.dumps({"workload": "patched training iteration + train.py's per-iteration extras (train.py:96, :113-116)",
                          "results": [run("patched", a.steps, a.warmup, extras=e) for e in ("masked", "patched")]}))
    else:
        print(json.dumps(run_both(a.steps, a.warmup)))
