"""Generates tests/golden/vae_small.npz by importing the reference's first-stage autoencoder modules
(svd_inpaint1/sgm/modules/diffusionmodules/model.py, sgm/modules/autoencoding/temporal_ae.py, the
DiagonalGaussianRegularizer) with the namespace-stub recipe of SURVEY.md Appendix A. Run ONLY in the build
container; /root/reference does not travel. The fixture holds the reference's OUTPUTS for seeded inputs and
weights (tests/svd_helpers.py regenerates both from the seeds).

Usage: python tools/gen_golden_vae.py
"""
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

from sgm.modules.diffusionmodules.model import Decoder, Encoder                              # noqa: E402
from sgm.modules.autoencoding.temporal_ae import VideoBlock, VideoDecoder                                # noqa: E402
from sgm.modules.autoencoding.regularizers import DiagonalGaussianRegularizer                # noqa: E402

out = {}
T = H.VAE_T
x = H.vae_inputs(31)

enc = Encoder(**H.SMALL_VAE).eval()
enc.load_state_dict(H.seeded_state_dict(enc, 41), strict=True)
out["enc_keys"] = np.array(sorted(enc.state_dict().keys()))
with torch.no_grad():
    moments = enc(x)
    out["enc_moments"] = moments.numpy()
    torch.manual_seed(H.VAE_SAMPLE_SEED)
    z, log = DiagonalGaussianRegularizer()(moments)
    out["z_sample"] = z.numpy()
    out["kl_loss"] = np.float64(log["kl_loss"])
    zm, _ = DiagonalGaussianRegularizer(sample=False)(moments)
    out["z_mode"] = zm.numpy()

    # time_mode "all" / "attn-only" cannot be constructed in the reference itself (temporal_ae.py:324 wraps the
    # FUNCTION make_time_attn in partialclass -> TypeError), so only the shipped "conv-only" mode has a golden.
    for mode in ("conv-only",):
        dec = VideoDecoder(**H.SMALL_VAE, video_kernel_size=[3, 1, 1], time_mode=mode).eval()
        dec.load_state_dict(H.seeded_state_dict(dec, 42), strict=True)
        tag = mode.replace("-", "_")
        out[f"vdec_keys_{tag}"] = np.array(sorted(dec.state_dict().keys()))
        acts = {}
        hooks = [dec.mid.block_1.register_forward_hook(lambda m, i, o: acts.__setitem__("mid1", o)),
                 dec.mid.attn_1.register_forward_hook(lambda m, i, o: acts.__setitem__("attn", o)),
                 dec.up[1].register_forward_hook(lambda m, i, o: None)]
        out[f"vdec_out_{tag}"] = dec(z, timesteps=T).numpy()
        for h in hooks:
            h.remove()
        for k, v in acts.items():
            out[f"vdec_act_{tag}_{k}"] = v.numpy()
        # two "videos" in one batch (en_and_decode_n_samples_a_time smaller than the batch never mixes videos,
        # but a caller may pass b > 1)
        z2 = torch.cat([z, z.flip(0)])
        out[f"vdec_out2_{tag}"] = dec(z2, timesteps=T).numpy()
        if mode == "conv-only":
            out["vdec_out_skip_video"] = dec(z, timesteps=T, skip_video=True).numpy()

    # the temporal attention block on its own (constructible directly)
    vb = VideoBlock(64).eval()
    vb.load_state_dict(H.seeded_state_dict(vb, 44), strict=True)
    out["vblock_keys"] = np.array(sorted(vb.state_dict().keys()))
    xb = torch.randn(2 * T, 64, 8, 4, generator=torch.Generator().manual_seed(32))
    out["vblock_out"] = vb(xb, timesteps=T).numpy()
    out["vblock_out_skip"] = vb(xb, timesteps=T, skip_video=True).numpy()

    pdec = Decoder(**H.SMALL_VAE).eval()
    pdec.load_state_dict(H.seeded_state_dict(pdec, 43), strict=True)
    out["dec_keys"] = np.array(sorted(pdec.state_dict().keys()))
    out["dec_out"] = pdec(z).numpy()

path = os.path.join(HERE, "..", "tests", "golden", "vae_small.npz")
np.savez_compressed(path, **out)
print("wrote", os.path.normpath(path), f"{os.path.getsize(path) / 1e6:.2f} MB;",
      {k: float(np.abs(v).mean()) for k, v in out.items() if v.dtype.kind == "f" and v.ndim > 0})
