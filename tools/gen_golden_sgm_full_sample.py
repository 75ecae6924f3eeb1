"""Adds to tests/golden/sgm_full.npz: the FIRST TWO STEPS of the reference's sampling loop at the full size of BASELINE.json
configs[3] — EulerEDMSampler(num_steps = 2, sigma_max 700) with the per-frame LinearPredictionGuider (1.0 -> 2.5, control_hint as
an additional condition key) over Denoiser(VScalingWithEDMcNoise) over ControlNet + ControlledVideoUNet (the apply_model of
models/csvd.py:1086-1152), 14 frames on the 72 x 128 latent (the guider doubles the batch to 28) — on the CPU in fp32, and the
error of the same loop under bf16 / f16 autocast against it as the budgets. Recorded: the sample after the two steps,
subsampled [:, :, ::2, ::2]. Same seeded weights as tools/gen_golden_sgm_full.py (run that first). Build container only
(~40 minutes on 8 cores).

Usage: python tools/gen_golden_sgm_full_sample.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, HERE)
import svd_helpers as H  # noqa: E402
from ref_import import import_reference  # noqa: E402

t0 = time.time()
log = lambda *a: print(f"[{time.time() - t0:7.1f} s]", *a, flush=True)
ref = import_reference()
from sgm.modules.diffusionmodules.denoiser import Denoiser  # noqa: E402
from sgm.modules.diffusionmodules.sampling import EulerEDMSampler  # noqa: E402

torch.manual_seed(0)
T = H.FULL_T
cunet = ref["ControlledVideoUNet"](**H.FULL_UNET).eval()
cunet.load_state_dict(H.seeded_state_dict(cunet, 71), strict=True)
cnet = ref["ControlNet"](**H.FULL_CTRL).eval()
cnet.load_state_dict(H.seeded_state_dict(cnet, 72), strict=True)
log("networks built")

one = H.seeded_inputs(74, T=T, hw=H.FULL_HW, cfg=H.FULL_UNET, cfg_doubled=False)
kw = dict(num_video_frames=T, image_only_indicator=one["image_only_indicator"])
den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
c = dict(crossattn=one["crossattn"], vector=one["vector"], concat=one["concat"], control_hint=one["control_hint"])
uc = dict(crossattn=torch.zeros_like(one["crossattn"]), vector=torch.zeros_like(one["vector"]),
          concat=torch.zeros_like(one["concat"]), control_hint=one["control_hint"])


def apply_model(x, t, cond_, num_video_frames=None, image_only_indicator=None):
    xi = torch.cat([x, cond_["concat"]], 1)
    cs = cnet(x=xi, hint=cond_["control_hint"], timesteps=t, context=cond_["crossattn"], y=cond_["vector"],
              num_video_frames=num_video_frames, image_only_indicator=image_only_indicator)
    return cunet(x=xi, timesteps=t, context=cond_["crossattn"], y=cond_["vector"], control=cs,
                 num_video_frames=num_video_frames, image_only_indicator=image_only_indicator)


def sample(tag):
    sampler = EulerEDMSampler(
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
        num_steps=H.FULL_SAMPLE_STEPS, device="cpu",
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"max_scale": 2.5, "min_scale": 1.0, "num_frames": T, "additional_cond_keys": ["control_hint"]}})
    n = [0]

    def denoiser(x, sigma, cc):
        d = den(apply_model, x, sigma, cc, **kw)
        n[0] += 1
        log(tag, "denoiser call", n[0])
        return d
    xs = sampler(denoiser, one["x"].clone(), c, uc=uc)
    return xs.float()[:, :, ::2, ::2].contiguous().numpy()


def rel_err(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    d = a - b
    return np.array([np.abs(d).max() / np.abs(b).max(), np.sqrt((d ** 2).mean()) / np.sqrt((b ** 2).mean())])


path = os.path.join(HERE, "..", "tests", "golden", "sgm_full.npz")
G = dict(np.load(path))
with torch.no_grad():
    f32 = sample("f32")
    with torch.autocast("cpu", dtype=torch.bfloat16):
        b16 = sample("bf16ac")
    with torch.autocast("cpu", dtype=torch.float16):
        f16 = sample("f16ac")
G["sample_final_f32"] = f32
G["budget_sample_final"] = rel_err(b16, f32)
G["budget_f16_sample_final"] = rel_err(f16, f32)
np.savez_compressed(path, **G)
log("added sample_final", f32.shape, "budgets bf16", G["budget_sample_final"], "f16", G["budget_f16_sample_final"], "mean|final|", float(np.abs(f32).mean()),
    f"{os.path.getsize(path) / 1e6:.2f} MB")
