"""Generates tests/golden/sgm_small.npz by importing the reference's denoise-loop modules
(svd_inpaint1/sgm/..., svd_inpaint1/models/csvd.py) with the namespace-stub recipe of SURVEY.md
Appendix A. Run ONLY in the build container; /root/reference does not travel. The fixture holds the
reference's OUTPUTS for seeded inputs/weights (tests/svd_helpers.py regenerates both from the seeds).

Usage: python tools/gen_golden_sgm.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import svd_helpers as H  # noqa: E402

ROOT = "/root/reference/svd_inpaint1"


def ns(name, path=None, **attrs):
    m = types.ModuleType(name)
    m.__package__ = name
    if path:
        m.__path__ = [path]
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


ns("sgm", ROOT + "/sgm")
ns("sgm.modules", ROOT + "/sgm/modules", UNCONDITIONAL_CONFIG={})
ns("sgm.modules.diffusionmodules", ROOT + "/sgm/modules/diffusionmodules")
ns("omegaconf", ListConfig=list, OmegaConf=dict)
ns("torchvision")
ns("pytorch_lightning", LightningModule=nn.Module)
ns("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
ns("sgm.models")
ns("sgm.models.diffusion", DiffusionEngine=type("DiffusionEngine", (nn.Module,), {}))
try:
    import matplotlib  # noqa: F401
except ImportError:
    ns("matplotlib", pyplot=types.ModuleType("pyplot"))
    ns("matplotlib.pyplot")
sys.path.insert(0, ROOT)

from sgm.modules.diffusionmodules.video_model import VideoUNet                      # noqa: E402
from sgm.modules.diffusionmodules.denoiser import Denoiser                          # noqa: E402
from sgm.modules.diffusionmodules.discretizer import EDMDiscretization              # noqa: E402
from sgm.modules.diffusionmodules.denoiser_scaling import VScalingWithEDMcNoise     # noqa: E402
from sgm.modules.diffusionmodules.guiders import LinearPredictionGuider             # noqa: E402
from sgm.modules.diffusionmodules.sampling import EulerEDMSampler                   # noqa: E402
from sgm.modules.diffusionmodules.util import timestep_embedding                    # noqa: E402
from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper                     # noqa: E402
csvd = importlib.import_module("models.csvd")

torch.manual_seed(0)
out = {}
T = H.T_FRAMES

# ---- scalar known-answer tests
disc = EDMDiscretization(sigma_min=0.002, sigma_max=700.0, rho=7.0)
out["sigmas25"] = disc(25).numpy()
sc = VScalingWithEDMcNoise()
sig = torch.tensor([700.0, 1.5, 0.002, 15.589973])
out["scaling_sigma"] = sig.numpy()
out["scaling_out"] = torch.stack(sc(sig)).numpy()
out["guider_scale14"] = LinearPredictionGuider(max_scale=2.5, num_frames=14, min_scale=1.0).scale.numpy()
out["temb_320"] = timestep_embedding(torch.tensor([0.25 * np.log(700.0), 0.0, -1.3]).float(), 320).numpy()
out["temb_odd"] = timestep_embedding(torch.tensor([3.0, 0.5]), 33, max_period=100).numpy()

# ---- networks (small config, every parameter re-randomised from a seed)
unet = VideoUNet(**H.SMALL_UNET).eval()
unet.load_state_dict(H.seeded_state_dict(unet, 11), strict=True)
cunet = csvd.ControlledVideoUNet(**H.SMALL_UNET).eval()
cunet.load_state_dict(H.seeded_state_dict(cunet, 11), strict=True)
cnet = csvd.ControlNet(**H.SMALL_CTRL).eval()
cnet.load_state_dict(H.seeded_state_dict(cnet, 12), strict=True)
out["unet_keys"] = np.array(sorted(unet.state_dict().keys()))
out["cnet_keys"] = np.array(sorted(cnet.state_dict().keys()))

inp = H.seeded_inputs(21)
kw = dict(num_video_frames=T, image_only_indicator=inp["image_only_indicator"])
with torch.no_grad():
    xin = torch.cat([inp["x"], inp["concat"]], 1)
    tt = 0.25 * inp["sigma"].log()
    # per-block activations through hooks
    acts = {}
    hooks = [unet.input_blocks[1].register_forward_hook(lambda m, i, o: acts.__setitem__("in1", o)),
             unet.input_blocks[3].register_forward_hook(lambda m, i, o: acts.__setitem__("in3", o)),
             unet.middle_block.register_forward_hook(lambda m, i, o: acts.__setitem__("mid", o)),
             unet.output_blocks[0].register_forward_hook(lambda m, i, o: acts.__setitem__("out0", o))]
    y_unet = unet(xin, tt, inp["crossattn"], inp["vector"], **kw)
    for h in hooks:
        h.remove()
    out["unet_out"] = y_unet.numpy()
    for k, v in acts.items():
        out["unet_act_" + k] = v.numpy()
    ctrls = cnet(xin, inp["control_hint"], tt, inp["crossattn"], inp["vector"], **kw)
    for i, c in enumerate(ctrls):
        out[f"ctrl_{i}"] = c.numpy()
    out["cunet_out"] = cunet(xin, tt, inp["crossattn"], inp["vector"], control=[c.clone() for c in ctrls], **kw).numpy()
    # image-only frames: alpha = 1 -> pure spatial path
    ind1 = torch.ones(1, T)
    out["unet_out_imageonly"] = unet(xin, tt, inp["crossattn"], inp["vector"], num_video_frames=T, image_only_indicator=ind1).numpy()

    # Denoiser.forward over OpenAIWrapper (stock path, diffusion.py:324-326)
    den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    wrap = OpenAIWrapper(unet)
    cond = dict(crossattn=inp["crossattn"], vector=inp["vector"], concat=inp["concat"])
    out["denoiser_out"] = den(wrap, inp["x"], inp["sigma"], cond, **kw).numpy()

    # 5-step Euler EDM trajectory with per-frame linear guidance, ControlNet path (csvd.py:1086-1152, :1258-1277)
    one = H.seeded_inputs(22, cfg_doubled=False)
    sampler = EulerEDMSampler(
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization",
                               "params": {"sigma_max": 700.0}},
        num_steps=5, device="cpu",
        guider_config={"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"max_scale": 2.5, "min_scale": 1.0, "num_frames": T,
                                  "additional_cond_keys": ["control_hint"]}})
    c = dict(crossattn=one["crossattn"], vector=one["vector"], concat=one["concat"], control_hint=one["control_hint"])
    uc = dict(crossattn=torch.zeros_like(one["crossattn"]), vector=torch.zeros_like(one["vector"]),
              concat=torch.zeros_like(one["concat"]), control_hint=one["control_hint"])

    def apply_model(x, t, cond_, num_video_frames=None, image_only_indicator=None):
        xi = torch.cat([x, cond_["concat"]], 1)
        cs = cnet(x=xi, hint=cond_["control_hint"], timesteps=t, context=cond_["crossattn"], y=cond_["vector"],
                  num_video_frames=num_video_frames, image_only_indicator=image_only_indicator)
        return cunet(x=xi, timesteps=t, context=cond_["crossattn"], y=cond_["vector"], control=cs,
                     num_video_frames=num_video_frames, image_only_indicator=image_only_indicator)
    traj = []

    def denoiser(x, sigma, cc):
        d = den(apply_model, x, sigma, cc, **kw)
        traj.append(d.clone())
        return d
    x0 = one["x"].clone()
    xs = sampler(denoiser, x0, c, uc=uc)
    out["sample_final"] = xs.numpy()
    out["sample_denoised_step0"] = traj[0].numpy()
    out["sample_denoised_step4"] = traj[4].numpy()

path = os.path.join(HERE, "..", "tests", "golden", "sgm_small.npz")
np.savez_compressed(path, **out)
print("wrote", os.path.normpath(path), f"{os.path.getsize(path) / 1e6:.2f} MB;",
      "unet params", sum(p.numel() for p in unet.parameters()), "mean|unet_out|", float(np.abs(out['unet_out']).mean()),
      "n_ctrl", len(ctrls), "mean|ctrl_last|", float(np.abs(out[f'ctrl_{len(ctrls)-1}']).mean()), "mean|final|", float(np.abs(out['sample_final']).mean()))
