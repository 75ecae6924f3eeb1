"""Generates tests/golden/vae_full.npz: the reference's first stage at its FULL size (configs/test/svd_f_est_ctrl_simp1.yaml:131-159)
on 576x1024 frames — the encoder (sgm/modules/diffusionmodules/model.py Encoder) on one frame and the video decoder
(sgm/modules/autoencoding/temporal_ae.py VideoDecoder, time_mode conv-only) on two latent frames, fp32 on the CPU as the reference
runs the first stage (disable_first_stage_autocast). Imported with the namespace-stub recipe of SURVEY.md Appendix A. Run ONLY in
the build container; /root/reference does not travel. Weights and inputs are regenerated from seeds on both sides
(tests/svd_helpers.py); the fixture holds the reference's OUTPUTS, subsampled (FULL_VAE_SUB) plus one dense crop.

Usage: python tools/gen_golden_vae_full.py        (about 3 minutes on 8 cores, ~12 GB)
"""
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

from sgm.modules.diffusionmodules.model import Encoder                                        # noqa: E402
from sgm.modules.autoencoding.temporal_ae import VideoDecoder                                 # noqa: E402

out = {}
T = H.FULL_VAE_T
with torch.no_grad():
    t0 = time.time()
    dec = VideoDecoder(**H.FULL_VAE, video_kernel_size=[3, 1, 1]).eval()
    dec.load_state_dict(H.seeded_state_dict(dec, 52), strict=True)
    out["vdec_keys"] = np.array(sorted(dec.state_dict().keys()))
    acts = {}
    hooks = [dec.mid.attn_1.register_forward_hook(lambda m, i, o: acts.__setitem__("mid_attn", o)),
             dec.up[1].block[2].register_forward_hook(lambda m, i, o: acts.__setitem__("up1", o))]
    y = dec(H.vae_full_latent(61), timesteps=T)
    for h in hooks:
        h.remove()
    assert tuple(y.shape) == (T, 3) + H.FULL_VAE_HW, y.shape
    out["vdec_out_sub"] = y[H.FULL_VAE_SUB].numpy().copy()
    out["vdec_out_crop"] = y[H.FULL_VAE_CROP].numpy().copy()
    out["vdec_out_absmax"] = np.float64(y.abs().max())
    out["vdec_out_mean"] = np.float64(y.double().mean())
    out["vdec_mid_attn_sub"] = acts["mid_attn"][:, ::8, ::4, ::4].numpy().copy()
    out["vdec_mid_attn_absmax"] = np.float64(acts["mid_attn"].abs().max())
    out["vdec_up1_sub"] = acts["up1"][:, ::16, ::8, ::8].numpy().copy()
    out["vdec_up1_absmax"] = np.float64(acts["up1"].abs().max())
    print(f"decoder {time.time() - t0:.0f} s", flush=True)
    del dec, y, acts

    t0 = time.time()
    enc = Encoder(**H.FULL_VAE).eval()
    enc.load_state_dict(H.seeded_state_dict(enc, 51), strict=True)
    out["enc_keys"] = np.array(sorted(enc.state_dict().keys()))
    x = H.vae_inputs(62, T=1, hw=H.FULL_VAE_HW)
    out["enc_moments"] = enc(x).numpy()
    print(f"encoder {time.time() - t0:.0f} s", flush=True)

path = os.path.join(HERE, "..", "tests", "golden", "vae_full.npz")
np.savez_compressed(path, **out)
print("wrote", os.path.normpath(path), f"{os.path.getsize(path) / 1e6:.2f} MB;",
      {k: float(np.abs(v).mean()) for k, v in out.items() if v.dtype.kind == "f" and v.ndim > 0})
