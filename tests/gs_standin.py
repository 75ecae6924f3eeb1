"""A stand-in of the reference's GaussianModel for the GPU tests of the training loop (the reference's scene/gaussian_model.py
cannot be imported here: plyfile is absent): the attributes, method names and BEHAVIOUR of gs-simp/scene/gaussian_model.py that
train.py / inpaint_rec.py and multiview_inpaint_amd.train_views rely on — stored parameters and activations (:44-59, :95-118), the
Adam groups (:154-163), reset_opacity / replace_tensor_to_optimizer (:297-333), prune (:335-382), cat_tensors_to_optimizer /
densification_postfix (:384-424), densify_and_split / densify_and_clone / densify_and_prune (:426-480), add_densification_stats
(:482-484) — written from that behaviour for these tests, on plain PyTorch ops. The methods patch_gs_simp swaps (prune_points,
cat_tensors_to_optimizer, add_densification_stats) are looked up on the instance's class, so a test can install the hooks on it."""
import math

import torch
from torch import nn


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def quaternion_to_matrix(q):
    """(w, x, y, z), normalised first — utils/general_utils.py:80-101."""
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.zeros((q.shape[0], 3, 3), device=q.device)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - r * z)
    R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y)
    R[:, 2, 1] = 2 * (y * z + r * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


class StandinGaussianModel:
    NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
    ATTRS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")

    def __init__(self, sc, max_deg, active_deg=None, dev="cuda", lrs=None, optimizer_cls=None, percent_dense=0.01):
        t = {k: torch.tensor(v, device=dev) for k, v in sc.items() if k != "sh_degree"}
        self.max_sh_degree = max_deg
        self.active_sh_degree = max_deg if active_deg is None else active_deg
        self.scaling_activation, self.opacity_activation = torch.exp, torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize
        self.percent_dense = percent_dense
        P = t["means3D"].shape[0]
        self._xyz = nn.Parameter(t["means3D"].clone())
        self._features_dc = nn.Parameter(t["shs"][:, :1].contiguous())
        self._features_rest = nn.Parameter(t["shs"][:, 1:].contiguous())
        self._opacity = nn.Parameter(inverse_sigmoid(t["opacities"].clamp(1e-4, 1 - 1e-4)))
        self._scaling = nn.Parameter(torch.log(t["scales"]))
        self._rotation = nn.Parameter(t["rotations"].clone())
        self.xyz_gradient_accum = torch.zeros(P, 1, device=dev)
        self.denom = torch.zeros(P, 1, device=dev)
        self.max_radii2D = torch.zeros(P, device=dev)
        self.lrs = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=2.5e-3 / 20, opacity=0.05, scaling=5e-3, rotation=1e-3)
        self.lrs.update(lrs or {})
        cls = optimizer_cls or torch.optim.Adam
        self.optimizer = cls([{"params": [getattr(self, a)], "lr": self.lrs[n], "name": n} for a, n in zip(self.ATTRS, self.NAMES)],
                             lr=0.0, eps=1e-15)

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: s.scaling_activation(s._scaling))
    get_rotation = property(lambda s: s.rotation_activation(s._rotation))
    get_opacity = property(lambda s: s.opacity_activation(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))

    def update_learning_rate(self, iteration):                  # (:165-171; a constant schedule is enough here)
        for g in self.optimizer.param_groups:
            if g["name"] == "xyz":
                g["lr"] = self.lrs["xyz"]
                return g["lr"]

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---- optimizer surgery ------------------------------------------------------------------------------------------------
    def replace_tensor_to_optimizer(self, tensor, name):
        out = {}
        for g in self.optimizer.param_groups:
            if g["name"] != name:
                continue
            st = self.optimizer.state.get(g["params"][0], None)
            if st is not None:
                st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(tensor), torch.zeros_like(tensor)
                del self.optimizer.state[g["params"][0]]
            g["params"][0] = nn.Parameter(tensor.requires_grad_(True))
            if st is not None:
                self.optimizer.state[g["params"][0]] = st
            out[name] = g["params"][0]
        return out

    def reset_opacity(self):
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        self._opacity = self.replace_tensor_to_optimizer(new, "opacity")["opacity"]

    def _prune_optimizer(self, keep):
        out = {}
        for g in self.optimizer.param_groups:
            st = self.optimizer.state.get(g["params"][0], None)
            if st is not None:
                st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][keep], st["exp_avg_sq"][keep]
                del self.optimizer.state[g["params"][0]]
            g["params"][0] = nn.Parameter(g["params"][0][keep].requires_grad_(True))
            if st is not None:
                self.optimizer.state[g["params"][0]] = st
            out[g["name"]] = g["params"][0]
        return out

    def prune_points(self, mask):
        keep = ~mask
        new = self._prune_optimizer(keep)
        for a, n in zip(self.ATTRS, self.NAMES):
            setattr(self, a, new[n])
        self.xyz_gradient_accum, self.denom, self.max_radii2D = self.xyz_gradient_accum[keep], self.denom[keep], self.max_radii2D[keep]

    def cat_tensors_to_optimizer(self, tensors_dict):
        out = {}
        for g in self.optimizer.param_groups:
            ext = tensors_dict[g["name"]]
            st = self.optimizer.state.get(g["params"][0], None)
            if st is not None:
                st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
                st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
                del self.optimizer.state[g["params"][0]]
            g["params"][0] = nn.Parameter(torch.cat((g["params"][0], ext), dim=0).requires_grad_(True))
            if st is not None:
                self.optimizer.state[g["params"][0]] = st
            out[g["name"]] = g["params"][0]
        return out

    def densification_postfix(self, new_xyz, new_dc, new_rest, new_opacity, new_scaling, new_rotation):
        new = self.cat_tensors_to_optimizer(dict(xyz=new_xyz, f_dc=new_dc, f_rest=new_rest, opacity=new_opacity, scaling=new_scaling,
                                                 rotation=new_rotation))
        for a, n in zip(self.ATTRS, self.NAMES):
            setattr(self, a, new[n])
        P, dev = self._xyz.shape[0], self._xyz.device
        self.xyz_gradient_accum, self.denom = torch.zeros(P, 1, device=dev), torch.zeros(P, 1, device=dev)
        self.max_radii2D = torch.zeros(P, device=dev)

    # ---- densification ----------------------------------------------------------------------------------------------------
    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2):
        P, dev = self._xyz.shape[0], self._xyz.device
        padded = torch.zeros(P, device=dev)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = (padded >= grad_threshold) & (self.get_scaling.max(dim=1).values > self.percent_dense * scene_extent)
        stds = self.get_scaling[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=dev), std=stds)
        rots = quaternion_to_matrix(self._rotation[sel]).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[sel].repeat(N, 1)
        new_scaling = torch.log(self.get_scaling[sel].repeat(N, 1) / (0.8 * N))
        self.densification_postfix(new_xyz, self._features_dc[sel].repeat(N, 1, 1), self._features_rest[sel].repeat(N, 1, 1),
                                   self._opacity[sel].repeat(N, 1), new_scaling, self._rotation[sel].repeat(N, 1))
        self.prune_points(torch.cat((sel, torch.zeros(N * int(sel.sum()), device=dev, dtype=torch.bool))))

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        sel = (torch.norm(grads, dim=-1) >= grad_threshold) & (self.get_scaling.max(dim=1).values <= self.percent_dense * scene_extent)
        self.densification_postfix(self._xyz[sel], self._features_dc[sel], self._features_rest[sel], self._opacity[sel],
                                   self._scaling[sel], self._rotation[sel])

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size):
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent)
        prune = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            prune = prune | (self.max_radii2D > max_screen_size) | (self.get_scaling.max(dim=1).values > 0.1 * extent)
        self.prune_points(prune)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1

    def params(self):
        return {n: getattr(self, a) for a, n in zip(self.ATTRS, self.NAMES)}


class StandinCamera:
    def __init__(self, cam, original_image, mask=None, inpainted=True, name=None, dev="cuda"):
        self.image_height, self.image_width = cam["H"], cam["W"]
        self.FoVx, self.FoVy = 2 * math.atan(cam["tanfovx"]), 2 * math.atan(cam["tanfovy"])
        self.world_view_transform = torch.tensor(cam["viewmatrix"], device=dev)
        self.full_proj_transform = torch.tensor(cam["projmatrix"], device=dev)
        self.camera_center = torch.tensor(cam["campos"], device=dev)
        self.original_image, self.mask, self.inpainted, self.image_name = original_image, mask, inpainted, name


class StandinOpt:
    """The fields of arguments/__init__.py:OptimizationParams the loop reads, with a densification schedule short enough for a test."""

    def __init__(self, **kw):
        self.iterations = 30
        self.lambda_dssim = 0.2
        self.densify_from_iter, self.densify_until_iter, self.densification_interval = 3, 20, 4
        self.opacity_reset_interval = 3000
        self.densify_grad_threshold = 0.0002
        self.random_background = False
        for k, v in kw.items():
            setattr(self, k, v)
