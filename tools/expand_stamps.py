"""Where a block of the binning partition kernel (csrc/raster_binning2.hip, expand_scatter_kernel) spends its cycles:
shader-clock stamps at the phase boundaries of every block, through the diagnostics hook mvi_raster_dev_stamps.
Run on the MI355X box: python tools/expand_stamps.py [N W H]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_inpaint_amd import _lib, raster as R, synthetic as syn  # noqa: E402

N, W, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1_500_000, 1920, 1080)
L = _lib.lib()
cam = syn.make_camera(W, H, 50.0)
sc = syn.make_scene(N, cam, 3, seed=0)
t = {k: torch.tensor(v, device="cuda") for k, v in sc.items() if k != "sh_degree"}
d = "cuda"
rs = R.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                                     bg=torch.zeros(3, device=d), scale_modifier=1.0,
                                     viewmatrix=torch.tensor(cam["viewmatrix"], device=d),
                                     projmatrix=torch.tensor(cam["projmatrix"], device=d), sh_degree=3,
                                     campos=torch.tensor(cam["campos"], device=d), prefiltered=False)
kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
for _ in range(3):
    R.rasterize_forward(rs, t["means3D"], t["opacities"], **kw)
torch.cuda.synchronize()
names = ["items requested", "masks cleared", "A masks built", "counts + layout scan", "bin starts", "C bit walk", "D write-out"]
for p in (1, 2):
    buf = torch.zeros(20000, 8, dtype=torch.int64, device=d)
    L.mvi_raster_dev_stamps(p, buf.data_ptr())
    R.rasterize_forward(rs, t["means3D"], t["opacities"], **kw)
    torch.cuda.synchronize()
    L.mvi_raster_dev_stamps(p, None)
    s = buf.cpu().numpy()
    s = s[(s[:, 0] != 0) & (s[:, 6] != 0)]
    dt = np.diff(s[:, :7], axis=1)
    print(f"pass {p}: {len(s)} blocks with entries; kernel span {(s[:, 6].max() - s[:, 0].min())} ticks (100 MHz memtime? see below)")
    for i, n in enumerate(names[1:]):
        print(f"   {n:28s} mean {dt[:, i].mean():9.0f}  p90 {np.percentile(dt[:, i], 90):9.0f}")
    print(f"   {'block total':28s} mean {(s[:, 6] - s[:, 0]).mean():9.0f}")
