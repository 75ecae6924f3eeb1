"""Generates tests/golden/raster_partial.npz by importing the reference's in-tree partial oracles
for the rasterizer path (run ONLY in the build container; /root/reference does not travel):

  * eval_sh                    gs-simp/utils/sh_utils.py:57-112  (+0.5, clamp_min 0 as at
                               gs-simp/gaussian_renderer/__init__.py:77-78)
  * build_scaling_rotation,    gs-simp/utils/general_utils.py:66-112 and the covariance builder
    strip_symmetric            gs-simp/scene/gaussian_model.py:27-31
  * getWorld2View2,            gs-simp/utils/graphics_utils.py:38-71 and the Camera matrix
    getProjectionMatrix        algebra of gs-simp/scene/cameras.py:60-63

The fixture holds inputs and the reference's outputs only (data, no source).
Usage: python tools/gen_golden_raster.py
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference/gs-simp"
sys.path.insert(0, REF)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "raster_partial.npz")

# the reference hard-codes device="cuda" in these helpers; there is no GPU here
_zeros = torch.zeros
def _cpu_zeros(*a, **k):
    k.pop("device", None)
    return _zeros(*a, **k)
torch.zeros = _cpu_zeros

from utils.sh_utils import eval_sh                       # noqa: E402
from utils.general_utils import build_scaling_rotation, strip_symmetric  # noqa: E402
from utils.graphics_utils import getWorld2View2, getProjectionMatrix     # noqa: E402

rng = np.random.default_rng(1234)
out = {}

# --- SH colour: [P,3,16] coefficient layout as the reference's python path uses (:74-76)
P = 64
dirs = rng.normal(size=(P, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
sh = np.concatenate([rng.normal(0, 1, (P, 1, 3)), rng.normal(0, 0.4, (P, 15, 3))], 1).astype(np.float32)  # [P,M,3]
out["sh_dirs"] = dirs.astype(np.float32)
out["sh_coeffs"] = sh
for deg in range(4):
    shs_view = torch.tensor(sh).transpose(1, 2)          # [P,3,16]
    rgb = torch.clamp_min(eval_sh(deg, shs_view, torch.tensor(dirs.astype(np.float32))) + 0.5, 0.0)
    out[f"sh_rgb_deg{deg}"] = rgb.numpy()

# --- covariance from scaling / rotation
scales = np.exp(rng.normal(-2, 0.7, (P, 3))).astype(np.float32)
rots = rng.normal(size=(P, 4)).astype(np.float32)
rots /= np.linalg.norm(rots, axis=1, keepdims=True)
for mod in (1.0, 0.5):
    L = build_scaling_rotation(mod * torch.tensor(scales), torch.tensor(rots))
    cov = strip_symmetric(L @ L.transpose(1, 2))
    out[f"cov3D_mod{mod}"] = cov.numpy()
out["cov_scales"], out["cov_rots"] = scales, rots

# --- camera matrices
cams = []
for i in range(4):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    T = rng.normal(size=3)
    W, H = [(800, 800), (1920, 1080), (384, 512), (61, 47)][i]
    fovy = math.radians([50.0, 50.0, 35.0, 70.0][i])
    fovx = 2 * math.atan(math.tan(fovy / 2) * W / H)
    wv = torch.tensor(getWorld2View2(R, T)).transpose(0, 1)
    pj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
    full = (wv.unsqueeze(0).bmm(pj.unsqueeze(0))).squeeze(0)
    center = wv.inverse()[3, :3]
    out[f"cam{i}_R"], out[f"cam{i}_T"] = R, T
    out[f"cam{i}_WHfovy"] = np.array([W, H, math.degrees(fovy)])
    out[f"cam{i}_world_view"], out[f"cam{i}_full_proj"], out[f"cam{i}_center"] = wv.numpy(), full.numpy(), center.numpy()

np.savez_compressed(OUT, **out)
print("wrote", os.path.normpath(OUT), {k: v.shape for k, v in out.items() if k.startswith("sh_rgb")})
