"""Stand-in of gs-simp/scene: InpaintScene over the synthetic scene of the two-rank training test (tests/dist_gpu_worker.py) — six
views of which every third is a non-inpainted one with a mask; the camera getters the launcher calls (scene/__init__.py:415-453 in the
reference), cameras_extent, model_path, save()."""
import os

import numpy as np
import torch

import dist_gpu_worker as W
import gs_standin as GS
from multiview_inpaint_amd import raster as R, synthetic as syn

from .gaussian_model import GaussianModel, InpaintGaussianModel  # noqa: F401


class InpaintScene:
    def __init__(self, args, gaussians, N=12_000, Wd=256, Hd=160):
        dev = torch.device("cuda", 0)
        deg = gaussians.max_sh_degree
        sc = syn.make_scene(N, syn.make_camera(Wd, Hd, 50.0), deg, seed=0, log_scale_mean=np.log(0.04))
        gaussians.load_standin(sc, dev)
        self.gaussians, self.model_path, self.cameras_extent = gaussians, args.model_path, 6.0
        g = torch.Generator(dev).manual_seed(77)
        shs = torch.tensor(sc["shs"], device=dev)
        shs[:, 0] += 0.6 * torch.randn(N, 3, device=dev, generator=g)
        t0 = {k: torch.tensor(sc[k], device=dev) for k in ("means3D", "opacities", "scales", "rotations")}
        self.cams = []
        for k in range(6):
            cam = W.camera(syn, np, k, Wd, Hd)
            rs = R.GaussianRasterizationSettings(
                image_height=Hd, image_width=Wd, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
                scale_modifier=1.0, viewmatrix=torch.tensor(cam["viewmatrix"], device=dev), projmatrix=torch.tensor(cam["projmatrix"], device=dev),
                sh_degree=deg, campos=torch.tensor(cam["campos"], device=dev), prefiltered=False)
            img, _, _, _ = R.rasterize_forward(rs, t0["means3D"], t0["opacities"], shs=shs, scales=t0["scales"], rotations=t0["rotations"])
            mask = None
            if k % 3 == 2:
                mask = torch.zeros(1, Hd, Wd, device=dev)
                mask[:, Hd // 4:Hd // 2, Wd // 3:2 * Wd // 3] = 1.0
            self.cams.append(GS.StandinCamera(cam, img, mask=mask, inpainted=mask is None, name=f"v{k}", dev=dev))

    def getInpaintTrainCameras(self, n_mode, ctrl_id=-1):
        return list(self.cams)

    def getInpaintCameras(self, n_mode, ctrl_id=-1):
        return [c for c in self.cams if c.inpainted]

    def getTrainCameras(self):
        return list(self.cams)

    def getTestCameras(self):
        return self.cams[:2]

    def save(self, iteration, out_dir):
        d = os.path.join(out_dir, f"point_cloud/iteration_{iteration}")
        os.makedirs(d, exist_ok=True)
        torch.save({n: getattr(self.gaussians, a).detach().cpu() for a, n in zip(self.gaussians.ATTRS, self.gaussians.NAMES)},
                   os.path.join(d, "point_cloud.pt"))
