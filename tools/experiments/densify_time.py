"""densify_and_prune at P = 1.5 M: the reference's recipe (tests/gs_standin.py restates gaussian_model.py:384-480) as written against the
same recipe with dropin.patch_gs_simp's cat_tensors_to_optimizer + prune_points on the class. Milliseconds per call (one call = clone
+ split + two prunes: what train.py runs every 100 iterations); beside tools/experiments/prune_points_time.py."""
import os, sys, time
import numpy as np
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from multiview_inpaint_amd import synthetic as syn, train_ops as T
from multiview_inpaint_amd.dropin import patch_gs_simp
import gs_standin as GS

P = 1_500_000
sc = syn.make_scene(P, syn.make_camera(64, 64, 50.0), 3, seed=0)


class Hooked(GS.StandinGaussianModel):
    cat_tensors_to_optimizer = patch_gs_simp._make_cat_tensors(GS.StandinGaussianModel.cat_tensors_to_optimizer)
    prune_points = patch_gs_simp._make_prune_points(GS.StandinGaussianModel.prune_points)


for name, cls in (("reference recipe", GS.StandinGaussianModel), ("import hooks", Hooked)) * 2:
    m = cls(sc, 3, optimizer_cls=T.FusedAdam)
    g = torch.Generator("cuda").manual_seed(4)
    for p in m.params().values():
        p.grad = torch.randn(p.shape, device="cuda", generator=g)
    m.optimizer.step()
    m.optimizer.zero_grad(set_to_none=True)
    m.xyz_gradient_accum = torch.rand(P, 1, device="cuda", generator=g) * 2.2e-4        # ~10 % of the Gaussians above the threshold
    m.denom = torch.ones(P, 1, device="cuda")
    m.max_radii2D = torch.rand(P, device="cuda", generator=g) * 21
    torch.manual_seed(9); torch.cuda.manual_seed(9)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.densify_and_prune(0.0002, 0.005, 6.0, 20)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) * 1e3:.2f} ms  (P {P} -> {m._xyz.shape[0]})", flush=True)
    del m
