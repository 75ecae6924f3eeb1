"""Stand-in of gs-simp/scene/gaussian_model.py for the launcher test: tests/gs_standin.StandinGaussianModel behind the reference's
construction sequence (GaussianModel(sh_degree) -> the scene fills it -> training_setup(opt)), with the methods
dropin.patch_gs_simp looks up in THIS module's classes defined here so that install() finds and wraps them."""
import torch

import gs_standin as GS


class GaussianModel(GS.StandinGaussianModel):
    def __init__(self, sh_degree):
        self.max_sh_degree, self.active_sh_degree = sh_degree, 0
        self.optimizer = None

    def load_standin(self, sc, dev="cuda"):
        GS.StandinGaussianModel.__init__(self, sc, self.max_sh_degree, active_deg=self.max_sh_degree, dev=dev)
        self.optimizer = None

    def training_setup(self, training_args):
        self.percent_dense = training_args.percent_dense
        groups = [{"params": [getattr(self, a)], "lr": self.lrs[n], "name": n} for a, n in zip(self.ATTRS, self.NAMES)]
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)

    def capture(self):
        return tuple(getattr(self, a).detach().clone() for a in self.ATTRS) + (self.optimizer.state_dict(),)

    # (defined here, not only inherited: patch_gs_simp wraps what it finds in the __dict__ of this module's classes)
    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        return GS.StandinGaussianModel.add_densification_stats(self, viewspace_point_tensor, update_filter)

    def prune_points(self, mask):
        return GS.StandinGaussianModel.prune_points(self, mask)

    def cat_tensors_to_optimizer(self, tensors_dict):
        return GS.StandinGaussianModel.cat_tensors_to_optimizer(self, tensors_dict)


class InpaintGaussianModel(GaussianModel):
    pass
