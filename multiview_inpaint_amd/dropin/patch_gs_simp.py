"""Opt-in hook: an UNCHANGED gs-simp training script gets the fused ops around the rasterizer.

With `multiview_inpaint_amd/dropin` on PYTHONPATH the reference's train.py / inpaint_rec.py already run on the HIP rasterizer
(`diff_gaussian_rasterization`, `simple_knn` resolve to this package). The photometric loss, the optimizer step and their ~30
small PyTorch kernels then dominate the iteration: 14.7 ms against 1.8 ms with the fused ops (BENCH_r03 `train_iteration`),
and INTEGRATION.md asked for two one-line edits of the reference to get them. This module makes those edits at import time
instead, so no reference file changes:

    python -m multiview_inpaint_amd.dropin.patch_gs_simp /path/to/gs-simp/train.py -s <scene> ...     # runs the script patched

or, inside a launcher of your own, BEFORE the training script is imported / executed:

    from multiview_inpaint_amd.dropin import patch_gs_simp
    patch_gs_simp.install()

What install() does (each part can be switched off):
  * loss      utils.loss_utils.l1_loss / ssim (gs-simp/utils/loss_utils.py:17-41; imported by name at train.py:17,
              inpaint_rec.py:16) become multiview_inpaint_amd.train_ops.l1_loss / ssim: same names, signatures and values, one
              fused HIP kernel pair each instead of ~25 PyTorch kernels. The originals stay reachable as `_reference_l1_loss` /
              `_reference_ssim`.
  * optimizer every class of scene.gaussian_model with a `training_setup` (gaussian_model.py:149-167, :639-657) is wrapped:
              the `torch.optim.Adam(l, lr=0.0, eps=1e-15)` it creates (:163, :653) is replaced by train_ops.FusedAdam over the SAME
              param_groups (names, per-group learning rates) and the same state layout, so update_learning_rate, densification's
              edits of optimizer.state, capture() / restore() run unchanged; one launch per step instead of 6 x 3.
  * render    gaussian_renderer.render (gaussian_renderer/__init__.py:18-104; imported by name at train.py:16, inpaint_rec.py:16,
              render.py:17, ...) hands the model's STORED parameters (_xyz, _features_dc, _features_rest, _opacity, _scaling,
              _rotation) to GaussianRasterizer.forward_raw: exp / sigmoid / normalize / cat of gaussian_model.py:95-115 and their
              chain rule run inside the preprocess kernels instead of as ~12 PyTorch kernels over the 1.5 M rows (the 288 MB
              torch.cat of the SH rows alone is 0.15 ms each way). Same arguments, same returned dict. Calls the patched form
              cannot serve — override_color, pipe.convert_SHs_python, pipe.compute_cov3D_python, a model whose activations or
              getters are not the stock ones — go to the reference's own render (kept as `_reference_render`).
  * stats     GaussianModel.add_densification_stats (gaussian_model.py:482-484; called every iteration below densify_until_iter,
              train.py:116) without boolean-mask indexing: `accum[mask] += norm(grad[mask, :2])` and `denom[mask] += 1` are a
              nonzero() with a host read-back, two gathers and two index_put each; the patched form adds
              `where(mask, norm(grad[:, :2]), 0)` and `mask` to the whole arrays — the same values, three elementwise kernels, no
              synchronisation. Other filter types (index tensors) go to the reference's method.
  * surgery   GaussianModel.prune_points (gaussian_model.py:351-365, with _prune_optimizer :335-349; twice per densification): the 6
              parameters, their 12 Adam moments and the 3 per-Gaussian statistics are compacted with ONE scan of the mask and ONE gather
              launch (train_ops.prune_optimizer_state) instead of 21 boolean-index operations with a nonzero() read-back each —
              bit-identical tensors, optimizer.state re-keyed the same way. Masks that are not boolean GPU tensors, or an optimizer
              whose groups are not one parameter each, go to the reference's method. GaussianModel.cat_tensors_to_optimizer
              (gaussian_model.py:384-404; behind densification_postfix, twice per densification) grows the 6 parameters and their
              12 moments through train_ops.extend_optimizer_state (moments: allocate + copy + memset of the new rows instead of a
              zeros_like temporary and a concatenation each) — identical tensors, optimizer.state re-keyed the same way; a dict or an
              optimizer of another shape goes to the reference's method.
Nothing is patched that is not named here; a script that imported the loss functions before install() ran keeps the
reference's (install() must come first — the runner below guarantees it)."""
import functools
import importlib
import os
import runpy
import sys

_DROPIN = os.path.dirname(os.path.abspath(__file__))


def _swap_optimizer(model):
    import torch
    from multiview_inpaint_amd.train_ops import FusedAdam
    opt = getattr(model, "optimizer", None)
    if type(opt) is not torch.optim.Adam:
        return False
    if any(g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False) for g in opt.param_groups):
        return False                                  # not the configuration FusedAdam implements: leave it alone
    keep = ("params", "lr", "betas", "eps", "name")
    groups = [{k: v for k, v in g.items() if k in keep} for g in opt.param_groups]
    new = FusedAdam(groups, lr=opt.defaults["lr"], betas=opt.defaults["betas"], eps=opt.defaults["eps"])
    for p, st in opt.state.items():                   # (empty right after training_setup; kept for callers that re-run it)
        new.state[p] = st
    model.optimizer = new
    return True


def _stock_model(pc, base):
    """True if `pc` computes what forward_raw computes: the stock activations (gaussian_model.py:33-41) and the stock getters."""
    import torch
    if not isinstance(pc, base):
        return False
    if (getattr(pc, "scaling_activation", None) is not torch.exp or getattr(pc, "opacity_activation", None) is not torch.sigmoid or
            getattr(pc, "rotation_activation", None) is not torch.nn.functional.normalize):
        return False
    cls = type(pc)
    return all(getattr(cls, n, None) is getattr(base, n) for n in ("get_xyz", "get_scaling", "get_rotation", "get_opacity", "get_features"))


def _make_render(reference_render, base_model):
    import math

    @functools.wraps(reference_render)
    def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
        if (override_color is not None or getattr(pipe, "convert_SHs_python", False) or getattr(pipe, "compute_cov3D_python", False)
                or not _stock_model(pc, base_model) or not pc._xyz.is_cuda):
            return reference_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color)
        import torch
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        # the screen-space means exist to carry a gradient back to the caller (train.py:87, :116: add_densification_stats reads its .grad)
        screenspace_points = torch.zeros_like(pc._xyz, requires_grad=True) + 0
        if screenspace_points.requires_grad:              # (not under torch.no_grad(): render.py, the evaluation renders)
            screenspace_points.retain_grad()
        settings = GaussianRasterizationSettings(
            image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
            tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
            scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
            projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
            campos=viewpoint_camera.camera_center, prefiltered=False)
        image, radii, depth = GaussianRasterizer(raster_settings=settings).forward_raw(
            pc._xyz, screenspace_points, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation)
        return {"render": image, "depth": depth, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
                "radii": radii}
    render._mvi_patched = True
    return render


def _make_stats(reference_stats):
    @functools.wraps(reference_stats)
    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        import torch
        g = getattr(viewspace_point_tensor, "grad", None)
        acc, den = getattr(self, "xyz_gradient_accum", None), getattr(self, "denom", None)
        if (g is None or not torch.is_tensor(update_filter) or update_filter.dtype != torch.bool or update_filter.ndim != 1
                or not torch.is_tensor(acc) or not torch.is_tensor(den) or g.ndim != 2 or g.shape[0] != update_filter.shape[0]
                or tuple(acc.shape) != (g.shape[0], 1) or tuple(den.shape) != (g.shape[0], 1)):
            return reference_stats(self, viewspace_point_tensor, update_filter)
        m = update_filter[:, None]
        n = torch.norm(g[:, :2], dim=-1, keepdim=True)
        acc += torch.where(m, n, 0.0).to(acc.dtype)
        den += m.to(den.dtype)
    add_densification_stats._mvi_patched = True
    return add_densification_stats


def _make_prune_points(reference_prune_points):
    names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
    attrs = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
    stat_names = ("xyz_gradient_accum", "denom", "max_radii2D")

    @functools.wraps(reference_prune_points)
    def prune_points(self, mask):
        import torch
        opt = getattr(self, "optimizer", None)
        ok = (torch.is_tensor(mask) and mask.dtype == torch.bool and mask.ndim == 1 and mask.is_cuda and opt is not None
              and all(len(g["params"]) == 1 for g in opt.param_groups)
              and sorted(g.get("name") for g in opt.param_groups) == sorted(names)
              and all(torch.is_tensor(getattr(self, n, None)) and getattr(self, n).shape[0] == mask.shape[0]
                      and getattr(self, n).element_size() == 4 for n in stat_names))
        if not ok:
            return reference_prune_points(self, mask)
        from multiview_inpaint_amd.train_ops import prune_optimizer_state
        tensors, stats = prune_optimizer_state(opt, ~mask, extra=tuple(getattr(self, n) for n in stat_names))
        for a, n in zip(attrs, names):
            setattr(self, a, tensors[n])
        for n, t in zip(stat_names, stats):
            setattr(self, n, t)
    prune_points._mvi_patched = True
    return prune_points


def _make_cat_tensors(reference_cat_tensors):
    names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")

    @functools.wraps(reference_cat_tensors)
    def cat_tensors_to_optimizer(self, tensors_dict):
        import torch
        opt = getattr(self, "optimizer", None)
        ok = (opt is not None and isinstance(tensors_dict, dict) and all(len(g["params"]) == 1 for g in opt.param_groups)
              and sorted(g.get("name") for g in opt.param_groups) == sorted(names)
              and all(torch.is_tensor(tensors_dict.get(n)) and tensors_dict[n].device == g["params"][0].device
                      and tensors_dict[n].dtype == g["params"][0].dtype and tensors_dict[n].shape[1:] == g["params"][0].shape[1:]
                      for g in opt.param_groups for n in (g["name"],)))
        if not ok:
            return reference_cat_tensors(self, tensors_dict)
        from multiview_inpaint_amd.train_ops import extend_optimizer_state
        return extend_optimizer_state(opt, tensors_dict)
    cat_tensors_to_optimizer._mvi_patched = True
    return cat_tensors_to_optimizer


def install(loss=True, optimizer=True, render=True, stats=True, surgery=True):
    """Patches the gs-simp modules named above (they must be importable: the script's directory on sys.path). Returns the
    list of what was patched, for logging. Idempotent."""
    if _DROPIN not in sys.path:
        sys.path.insert(0, _DROPIN)                   # diff_gaussian_rasterization, simple_knn
    done = []
    if loss:
        from multiview_inpaint_amd import train_ops
        lu = importlib.import_module("utils.loss_utils")
        if getattr(lu, "_mvi_patched", False) is False:
            lu._reference_l1_loss, lu._reference_ssim = lu.l1_loss, lu.ssim
            lu.l1_loss, lu.ssim = train_ops.l1_loss, train_ops.ssim
            lu._mvi_patched = True
        done += ["utils.loss_utils.l1_loss", "utils.loss_utils.ssim"]
    if optimizer:
        gm = importlib.import_module("scene.gaussian_model")
        for name, cls in list(vars(gm).items()):
            setup = isinstance(cls, type) and cls.__dict__.get("training_setup")
            if not setup or getattr(setup, "_mvi_patched", False):
                continue

            def wrap(fn):
                @functools.wraps(fn)
                def training_setup(self, *a, **k):
                    out = fn(self, *a, **k)
                    _swap_optimizer(self)
                    return out
                training_setup._mvi_patched = True
                return training_setup
            cls.training_setup = wrap(setup)
            done.append(f"scene.gaussian_model.{name}.training_setup")
    if stats:
        gm = importlib.import_module("scene.gaussian_model")
        for name, cls in list(vars(gm).items()):
            fn = isinstance(cls, type) and cls.__dict__.get("add_densification_stats")
            if not fn or getattr(fn, "_mvi_patched", False):
                continue
            cls.add_densification_stats = _make_stats(fn)
            done.append(f"scene.gaussian_model.{name}.add_densification_stats")
    if surgery:
        gm = importlib.import_module("scene.gaussian_model")
        for name, cls in list(vars(gm).items()):
            fn = isinstance(cls, type) and cls.__dict__.get("prune_points")
            if not fn or getattr(fn, "_mvi_patched", False):
                continue
            cls.prune_points = _make_prune_points(fn)
            done.append(f"scene.gaussian_model.{name}.prune_points")
        for name, cls in list(vars(gm).items()):
            fn = isinstance(cls, type) and cls.__dict__.get("cat_tensors_to_optimizer")
            if not fn or getattr(fn, "_mvi_patched", False):
                continue
            cls.cat_tensors_to_optimizer = _make_cat_tensors(fn)
            done.append(f"scene.gaussian_model.{name}.cat_tensors_to_optimizer")
    if render:
        gr = importlib.import_module("gaussian_renderer")
        gm = importlib.import_module("scene.gaussian_model")
        if not getattr(gr.render, "_mvi_patched", False):
            gr._reference_render = gr.render
            gr.render = _make_render(gr.render, gm.GaussianModel)
        done.append("gaussian_renderer.render")
    return done


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    sys.path.insert(0, os.path.dirname(script))       # what `python script.py` does
    done = install()
    print("[multiview_inpaint_amd] patched: " + ", ".join(done), file=sys.stderr)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
