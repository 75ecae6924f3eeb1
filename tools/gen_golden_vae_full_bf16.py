"""Adds to tests/golden/vae_full.npz the error of the REFERENCE's own bf16-autocast decode (torch.autocast on the CPU around the
reference's VideoDecoder, same seeded weights and latents as tools/gen_golden_vae_full.py) against its fp32 decode: the budget a
reduced-precision first-stage decode of this package is held to (tests/test_vae_gpu.py, tools/bench_vae.py --dtype bf16). The
reference itself runs the first stage in fp32 (disable_first_stage_autocast: True, configs/test/svd_f_est_ctrl_simp1.yaml:6); the
bf16 decode is an opt-in of this package. Run ONLY in the build container. Usage: python tools/gen_golden_vae_full_bf16.py"""
import os
import sys
import time
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
ns("sgm.modules.autoencoding", ROOT + "/sgm/modules/autoencoding")
ns("sgm.modules.distributions", ROOT + "/sgm/modules/distributions")
ns("omegaconf", ListConfig=list, OmegaConf=dict)
ns("torchvision")
ns("pytorch_lightning", LightningModule=nn.Module)
ns("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
try:
    import matplotlib  # noqa: F401
except ImportError:
    ns("matplotlib", pyplot=types.ModuleType("pyplot"))
    ns("matplotlib.pyplot")
sys.path.insert(0, ROOT)

from sgm.modules.autoencoding.temporal_ae import VideoDecoder                                 # noqa: E402

path = os.path.join(HERE, "..", "tests", "golden", "vae_full.npz")
G = dict(np.load(path))
T = H.FULL_VAE_T
with torch.no_grad():
    t0 = time.time()
    dec = VideoDecoder(**H.FULL_VAE, video_kernel_size=[3, 1, 1]).eval()
    dec.load_state_dict(H.seeded_state_dict(dec, 52), strict=True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        y = dec(H.vae_full_latent(61), timesteps=T).float()
    print(f"bf16-autocast decode {time.time() - t0:.0f} s", flush=True)


def err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    d = (a - b).abs()
    return np.array([float(d.max()), float(d.pow(2).mean().sqrt())])


amax = float(G["vdec_out_absmax"])
G["budget_bf16_out_sub"] = err(y[H.FULL_VAE_SUB], G["vdec_out_sub"]) / amax          # (max, rms) of the error, relative to max |fp32 output|
G["budget_bf16_out_crop"] = err(y[H.FULL_VAE_CROP], G["vdec_out_crop"]) / amax
np.savez_compressed(path, **G)
print("wrote", os.path.normpath(path), {k: G[k].tolist() for k in ("budget_bf16_out_sub", "budget_bf16_out_crop")})
