"""Generates tests/golden/sgm_hd64.npz: the reference's VideoUNet / ControlNet / ControlledVideoUNet at the
PRODUCTION head width (num_head_channels 64, tests/svd_helpers.py:SMALL_UNET64) on a 16x16 latent, so that the build's
bf16 MFMA attention kernel (D = 64, S_k = 256 and 64) meets outputs the reference itself produced:
  * fp32 outputs (`*_f32`), and
  * the same modules under the reference's own reduced-precision recipe — autocast over fp32 weights
    (svd_inpaint1/models/csvd.py:27-31, configs/test/svd_f_est_ctrl_simp1.yaml:214) — with bf16 as the autocast type
    on the CPU (`*_bf16ac`): the error budget the build's bf16 path is held to.
Run ONLY in the build container. Fixture = reference OUTPUTS for seeded inputs/weights.

Usage: python tools/gen_golden_sgm_hd64.py
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

unet = ref["VideoUNet"](**H.SMALL_UNET64).eval()
unet.load_state_dict(H.seeded_state_dict(unet, 31), strict=True)
cunet = ref["ControlledVideoUNet"](**H.SMALL_UNET64).eval()
cunet.load_state_dict(H.seeded_state_dict(cunet, 31), strict=True)
cnet = ref["ControlNet"](**H.SMALL_CTRL64).eval()
cnet.load_state_dict(H.seeded_state_dict(cnet, 32), strict=True)

inp = H.seeded_inputs(41, hw=H.LATENT_HW64, cfg=H.SMALL_UNET64)
kw = dict(num_video_frames=T, image_only_indicator=inp["image_only_indicator"])
xin = torch.cat([inp["x"], inp["concat"]], 1)
tt = 0.25 * inp["sigma"].log()


def run(tag):
    y = unet(xin, tt, inp["crossattn"], inp["vector"], **kw)
    ctrls = cnet(xin, inp["control_hint"], tt, inp["crossattn"], inp["vector"], **kw)
    yc = cunet(xin, tt, inp["crossattn"], inp["vector"], control=[c.clone() for c in ctrls], **kw)
    out["unet_out_" + tag] = y.float().numpy()
    out["cunet_out_" + tag] = yc.float().numpy()
    for i, c in enumerate(ctrls):
        out[f"ctrl_{i}_{tag}"] = c.float().numpy()
    return len(ctrls)


with torch.no_grad():
    n = run("f32")
    with torch.autocast("cpu", dtype=torch.bfloat16):
        run("bf16ac")
out["n_ctrl"] = np.array(n)

# ---- the same three networks on a 32x32 latent (S = 1024 at level 0: the production 8-wave attention kernel's range).
# Only the two network outputs and the last control residual are kept (the others are 1.5 MB each); budgets for bf16 AND
# fp16 autocast (fp16 is the reference's own recipe on the GPU, configs/test/svd_f_est_ctrl_simp1.yaml:214).
inpL = H.seeded_inputs(43, hw=H.LATENT_HW64_L, cfg=H.SMALL_UNET64)
kwL = dict(num_video_frames=T, image_only_indicator=inpL["image_only_indicator"])
xinL = torch.cat([inpL["x"], inpL["concat"]], 1)
ttL = 0.25 * inpL["sigma"].log()


def runL(tag):
    y = unet(xinL, ttL, inpL["crossattn"], inpL["vector"], **kwL)
    ctrls = cnet(xinL, inpL["control_hint"], ttL, inpL["crossattn"], inpL["vector"], **kwL)
    yc = cunet(xinL, ttL, inpL["crossattn"], inpL["vector"], control=[c.clone() for c in ctrls], **kwL)
    out["L_unet_out_" + tag] = y.float().numpy()
    out["L_cunet_out_" + tag] = yc.float().numpy()
    out["L_ctrl_last_" + tag] = ctrls[-1].float().numpy()


with torch.no_grad():
    runL("f32")
    with torch.autocast("cpu", dtype=torch.bfloat16):
        runL("bf16ac")
    with torch.autocast("cpu", dtype=torch.float16):
        runL("f16ac")
path = os.path.join(HERE, "..", "tests", "golden", "sgm_hd64.npz")
np.savez_compressed(path, **out)
e = lambda a, b: float(np.abs(out[a] - out[b]).max() / np.abs(out[b]).max())
print("wrote", os.path.normpath(path), f"{os.path.getsize(path) / 1e6:.2f} MB; unet params",
      sum(p.numel() for p in unet.parameters()), "autocast-vs-fp32 rel: unet", e("unet_out_bf16ac", "unet_out_f32"),
      "cunet", e("cunet_out_bf16ac", "cunet_out_f32"), "mean|unet_out|", float(np.abs(out["unet_out_f32"]).mean()),
      "| 32x32: bf16ac", e("L_cunet_out_bf16ac", "L_cunet_out_f32"), "f16ac", e("L_cunet_out_f16ac", "L_cunet_out_f32"))
