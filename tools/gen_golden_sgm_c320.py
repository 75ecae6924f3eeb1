"""Generates tests/golden/sgm_c320.npz: the reference's VideoUNet / ControlNet / ControlledVideoUNet at the PRODUCTION channel
widths (model_channels 320, num_head_channels 64, tests/svd_helpers.py:SMALL_UNET320) on a 16x16 latent, so that the build's
implicit-GEMM convolutions (C_out = 320 / 640, 3x3 and (3,1,1)), the token-major VideoResBlock and the MFMA temporal attention
meet outputs the reference itself produced:
  * fp32 outputs (`*_f32`), and
  * the same modules under the reference's own reduced-precision recipe — autocast over fp32 weights
    (svd_inpaint1/models/csvd.py:27-31, configs/test/svd_f_est_ctrl_simp1.yaml:214) — in bf16 and f16 on the CPU
    (`*_bf16ac`, `*_f16ac`): the error budget the build's reduced-precision path is held to.
Run ONLY in the build container. Fixture = reference OUTPUTS for seeded inputs/weights (the two network outputs, the last
control residual and four subsampled intermediate block outputs of the UNet).

Usage: python tools/gen_golden_sgm_c320.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, HERE)
import svd_helpers as H  # noqa: E402
from ref_import import import_reference  # noqa: E402

ref = import_reference()
torch.manual_seed(0)
T = H.T_FRAMES
out = {}

unet = ref["VideoUNet"](**H.SMALL_UNET320).eval()
unet.load_state_dict(H.seeded_state_dict(unet, 51), strict=True)
cunet = ref["ControlledVideoUNet"](**H.SMALL_UNET320).eval()
cunet.load_state_dict(H.seeded_state_dict(cunet, 51), strict=True)
cnet = ref["ControlNet"](**H.SMALL_CTRL320).eval()
cnet.load_state_dict(H.seeded_state_dict(cnet, 52), strict=True)

inp = H.seeded_inputs(53, hw=H.LATENT_HW320, cfg=H.SMALL_UNET320)
inp["image_only_indicator"][0, 1] = 1.0              # one frame blended as an image (alpha = 1): the AlphaBlender's other branch
kw = dict(num_video_frames=T, image_only_indicator=inp["image_only_indicator"])
xin = torch.cat([inp["x"], inp["concat"]], 1)
tt = 0.25 * inp["sigma"].log()


# Intermediate block outputs of the plain UNet (round 4: a wrong block that the later layers wash out would pass a test of the
# final tensors only): the first level-0 block with its transformer, the first block of level 1, the middle block and the
# second-to-last output block, subsampled [:, ::4, ::2, ::2] (1/16 of the elements) to keep the fixture small.
PROBES = H.C320_PROBES
_seen = {}
for name in PROBES:
    mod = unet.get_submodule(name)
    mod.register_forward_hook(lambda m, i, o, name=name: _seen.__setitem__(name, o.detach().float()[:, ::4, ::2, ::2].contiguous().numpy()))


def run(tag):
    _seen.clear()
    y = unet(xin, tt, inp["crossattn"], inp["vector"], **kw)
    for name in PROBES:
        out[f"probe_{name}_{tag}"] = _seen[name]
    ctrls = cnet(xin, inp["control_hint"], tt, inp["crossattn"], inp["vector"], **kw)
    yc = cunet(xin, tt, inp["crossattn"], inp["vector"], control=[c.clone() for c in ctrls], **kw)
    out["unet_out_" + tag] = y.float().numpy()
    out["cunet_out_" + tag] = yc.float().numpy()
    out["ctrl_last_" + tag] = ctrls[-1].float().numpy()
    return len(ctrls)


with torch.no_grad():
    n = run("f32")
    with torch.autocast("cpu", dtype=torch.bfloat16):
        run("bf16ac")
    with torch.autocast("cpu", dtype=torch.float16):
        run("f16ac")
out["n_ctrl"] = np.array(n)
path = os.path.join(HERE, "..", "tests", "golden", "sgm_c320.npz")
np.savez_compressed(path, **out)
e = lambda a, b: float(np.abs(out[a] - out[b]).max() / np.abs(out[b]).max())
print("wrote", os.path.normpath(path), f"{os.path.getsize(path) / 1e6:.2f} MB; unet params", sum(p.numel() for p in unet.parameters()),
      "autocast-vs-fp32 rel: bf16 unet", e("unet_out_bf16ac", "unet_out_f32"), "cunet", e("cunet_out_bf16ac", "cunet_out_f32"),
      "f16 cunet", e("cunet_out_f16ac", "cunet_out_f32"), "mean|unet_out|", float(np.abs(out["unet_out_f32"]).mean()))
