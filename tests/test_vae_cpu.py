"""CPU parity of the first-stage autoencoder restatement (multiview_inpaint_amd.svd.vae, imported through the drop-in
`sgm` names) against golden outputs of the REFERENCE modules (tests/golden/vae_small.npz, made by
tools/gen_golden_vae.py from /root/reference). Tolerance 1e-4 relative; fp32 CPU results agree to ~1e-6."""
import os
import sys

import numpy as np
import pytest
import torch

import svd_helpers as H

DROPIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multiview_inpaint_amd", "dropin")
if DROPIN not in sys.path:
    sys.path.insert(0, DROPIN)

from sgm.util import instantiate_from_config  # noqa: E402

RTOL = 1e-4
# configs/test/svd_f_est_ctrl_simp1.yaml:124-159 with the small sizes
FIRST_STAGE = {"target": "sgm.models.autoencoder.AutoencodingEngine", "params": {
    "loss_config": {"target": "torch.nn.Identity"},
    "regularizer_config": {"target": "sgm.modules.autoencoding.regularizers.DiagonalGaussianRegularizer"},
    "encoder_config": {"target": "sgm.modules.diffusionmodules.model.Encoder", "params": H.SMALL_VAE},
    "decoder_config": {"target": "sgm.modules.autoencoding.temporal_ae.VideoDecoder",
                       "params": dict(H.SMALL_VAE, video_kernel_size=[3, 1, 1])}}}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "vae_small.npz"))


@pytest.fixture(scope="module")
def engine():
    eng = instantiate_from_config(FIRST_STAGE).eval()
    eng.encoder.load_state_dict(H.seeded_state_dict(eng.encoder, 41), strict=True)
    eng.decoder.load_state_dict(H.seeded_state_dict(eng.decoder, 42), strict=True)
    return eng


def test_state_dict_keys_match_the_reference(G, engine):
    from sgm.modules.autoencoding.temporal_ae import VideoBlock
    from sgm.modules.diffusionmodules.model import Decoder
    assert sorted(engine.encoder.state_dict().keys()) == list(G["enc_keys"])
    assert sorted(engine.decoder.state_dict().keys()) == list(G["vdec_keys_conv_only"])
    assert sorted(Decoder(**H.SMALL_VAE).state_dict().keys()) == list(G["dec_keys"])
    assert sorted(VideoBlock(64).state_dict().keys()) == list(G["vblock_keys"])
    assert sorted(k for k in engine.state_dict().keys()) == sorted(
        ["encoder." + k for k in G["enc_keys"]] + ["decoder." + k for k in G["vdec_keys_conv_only"]])


def test_encoder_and_regularizer(G, engine):
    x = H.vae_inputs(31)
    with torch.no_grad():
        m, _ = engine.encode(x, unregularized=True)
        assert rel(m, G["enc_moments"]) < RTOL
        torch.manual_seed(H.VAE_SAMPLE_SEED)
        z, log = engine.encode(x, return_reg_log=True)
    assert rel(z, G["z_sample"]) < RTOL                     # same CPU generator stream as the reference
    assert abs(float(log["kl_loss"]) - float(G["kl_loss"])) < 1e-4 * abs(float(G["kl_loss"]))
    from sgm.modules.autoencoding.regularizers import DiagonalGaussianRegularizer
    zm, _ = DiagonalGaussianRegularizer(sample=False)(torch.tensor(G["enc_moments"]))
    assert rel(zm, G["z_mode"]) < 1e-6


def test_video_decoder(G, engine):
    z = torch.tensor(G["z_sample"])
    acts = {}
    hooks = [engine.decoder.mid.block_1.register_forward_hook(lambda m, i, o: acts.__setitem__("mid1", o)),
             engine.decoder.mid.attn_1.register_forward_hook(lambda m, i, o: acts.__setitem__("attn", o))]
    with torch.no_grad():
        y = engine.decode(z, timesteps=H.VAE_T)
    for h in hooks:
        h.remove()
    assert rel(acts["mid1"], G["vdec_act_conv_only_mid1"]) < RTOL
    assert rel(acts["attn"], G["vdec_act_conv_only_attn"]) < RTOL
    assert rel(y, G["vdec_out_conv_only"]) < RTOL
    with torch.no_grad():
        assert rel(engine.decode(torch.cat([z, z.flip(0)]), timesteps=H.VAE_T), G["vdec_out2_conv_only"]) < RTOL
        assert rel(engine.decode(z, timesteps=H.VAE_T, skip_video=True), G["vdec_out_skip_video"]) < RTOL


def test_decode_first_stage_chunks(G, engine):
    """sgm/models/diffusion.py:193-212: a chunk of n frames is one video; chunk == batch reproduces decode()."""
    from multiview_inpaint_amd.svd.vae import decode_first_stage, encode_first_stage
    z = torch.tensor(G["z_sample"])
    y = decode_first_stage(engine, 0.18215 * z, scale_factor=0.18215)
    assert rel(y, G["vdec_out_conv_only"]) < RTOL
    y2 = decode_first_stage(engine, 0.18215 * torch.cat([z, z.flip(0)]), en_and_decode_n_samples_a_time=H.VAE_T)
    assert rel(y2[:H.VAE_T], G["vdec_out_conv_only"]) < RTOL
    torch.manual_seed(H.VAE_SAMPLE_SEED)
    assert rel(encode_first_stage(engine, H.vae_inputs(31)), 0.18215 * G["z_sample"]) < RTOL


def test_plain_decoder_and_video_block(G):
    from sgm.modules.autoencoding.temporal_ae import VideoBlock
    from sgm.modules.diffusionmodules.model import Decoder
    dec = Decoder(**H.SMALL_VAE).eval()
    dec.load_state_dict(H.seeded_state_dict(dec, 43), strict=True)
    with torch.no_grad():
        assert rel(dec(torch.tensor(G["z_sample"])), G["dec_out"]) < RTOL
    vb = VideoBlock(64).eval()
    vb.load_state_dict(H.seeded_state_dict(vb, 44), strict=True)
    xb = torch.randn(2 * H.VAE_T, 64, 8, 4, generator=torch.Generator().manual_seed(32))
    with torch.no_grad():
        assert rel(vb(xb, timesteps=H.VAE_T), G["vblock_out"]) < RTOL
        assert rel(vb(xb, timesteps=H.VAE_T, skip_video=True), G["vblock_out_skip"]) < RTOL


def test_full_size_parameter_count():
    """The shipped first-stage config (yaml :131-159) on the meta device: parameter counts of the reference modules
    (measured once with tools/gen_golden_vae.py's import recipe: Encoder 34 163 592, VideoDecoder 63 579 183)."""
    full = dict(attn_type="vanilla", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    from sgm.modules.autoencoding.temporal_ae import VideoDecoder
    from sgm.modules.diffusionmodules.model import Encoder
    with torch.device("meta"):
        enc, dec = Encoder(**full), VideoDecoder(**full, video_kernel_size=[3, 1, 1])
    assert sum(p.numel() for p in enc.parameters()) == 34163592
    assert sum(p.numel() for p in dec.parameters()) == 63579183


def test_full_size_state_dict_keys_match_the_reference(golden_dir):
    """The full-size first stage (svd_helpers.FULL_VAE) builds with exactly the reference's parameter names (construction only; the
    outputs are compared on the GPU, tests/test_vae_gpu.py)."""
    from sgm.modules.autoencoding.temporal_ae import VideoDecoder
    from sgm.modules.diffusionmodules.model import Encoder
    G = np.load(os.path.join(golden_dir, "vae_full.npz"))
    assert sorted(Encoder(**H.FULL_VAE).state_dict().keys()) == list(G["enc_keys"])
    assert sorted(VideoDecoder(**H.FULL_VAE, video_kernel_size=[3, 1, 1]).state_dict().keys()) == list(G["vdec_keys"])
    assert G["vdec_out_sub"].shape == (H.FULL_VAE_T, 3, H.FULL_VAE_HW[0] // 4, H.FULL_VAE_HW[1] // 4)
