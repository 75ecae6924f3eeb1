"""Shared by tools/gen_golden_sgm.py (reference side, build container only) and the sgm parity tests:
the small network configuration, seeded weights and seeded inputs. Weights are regenerated from the
seed on both sides (same state-dict keys and shapes by construction), so fixtures hold only inputs'
seeds and the reference's outputs."""
import os

import torch


def report(line):
    """print + append to the file named by MVI_PARITY_REPORT (the GPU runs set it: pytest -q shows no output of passing tests)."""
    print(line)
    path = os.environ.get("MVI_PARITY_REPORT")
    if path:
        with open(path, "a") as fh:
            fh.write(os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0] + " | " + line + "\n")


SMALL_UNET = dict(in_channels=8, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=[2, 1],
                  channel_mult=[1, 2], num_head_channels=16, transformer_depth=1, context_dim=24, adm_in_channels=12,
                  num_classes="sequential", use_linear_in_transformer=True, extra_ff_mix_layer=True,
                  use_spatial_context=True, merge_strategy="learned_with_images", video_kernel_size=[3, 1, 1],
                  use_checkpoint=False, spatial_transformer_attn_type="softmax")
SMALL_CTRL = {k: v for k, v in SMALL_UNET.items() if k != "out_channels"}
SMALL_CTRL["hint_channels"] = 7
T_FRAMES = 3
LATENT_HW = (16, 8)

# Second small configuration with the PRODUCTION head width (num_head_channels 64,
# configs/test/svd_f_est_ctrl_simp1.yaml:31): every spatial self-attention has D = 64 and S_k = 256 / 64 > 32,
# so in bf16/f16 it runs the MFMA flash kernel (mvi_attention_kernel_kind == 1) inside the module graph.
SMALL_UNET64 = dict(SMALL_UNET, model_channels=64, num_head_channels=64)
SMALL_CTRL64 = {k: v for k, v in SMALL_UNET64.items() if k != "out_channels"}
SMALL_CTRL64["hint_channels"] = 7
LATENT_HW64 = (16, 16)
# the same networks on a 32x32 latent: the level-0 spatial self-attention has S_q = S_k = 1024, the range of the PRODUCTION
# 8-wave kernel (csrc/attn_flash8.hip: S_q >= 1024 and S_k >= 256); level 1 (S = 256) stays on the 4-wave kernel
LATENT_HW64_L = (32, 32)

# Third small configuration with the PRODUCTION widths (model_channels 320 -> channels 320 / 640, num_head_channels 64,
# configs/test/svd_f_est_ctrl_simp1.yaml:18-31) on a 16x16 latent: the shapes the round-3 kernels are built for — the 3x3 and
# (3,1,1) convolutions as implicit GEMMs with C_out = 320 g (csrc/linear_n320.hip, K split at this image size), the token-major
# VideoResBlock, the temporal attention on MFMA (csrc/attn_temporal.hip, D = 64, H = 5 / 10), Upsample on tokens.
SMALL_UNET320 = dict(SMALL_UNET, model_channels=320, num_head_channels=64)
SMALL_CTRL320 = {k: v for k, v in SMALL_UNET320.items() if k != "out_channels"}
SMALL_CTRL320["hint_channels"] = 7
LATENT_HW320 = (16, 16)
# submodules of the UNet whose outputs tests/golden/sgm_c320.npz records (subsampled [:, ::4, ::2, ::2])
C320_PROBES = ("input_blocks.1", "input_blocks.3", "middle_block", "output_blocks.2")


# The FULL-SIZE configuration of BASELINE.json configs[3] (configs/test/svd_f_est_ctrl_simp1.yaml:19-61: model_channels 320,
# channel_mult [1, 2, 4, 4], 2 ResBlocks per level, attention at 4 / 2 / 1) on the 72x128 latent of 14 x 576x1024 frames, CFG batch 28
# — tests/golden/sgm_full.npz (tools/gen_golden_sgm_full.py). The reference side runs it with the plain softmax attention and
# without gradient checkpointing (the same arithmetic).
FULL_UNET = dict(in_channels=8, out_channels=4, model_channels=320, channel_mult=[1, 2, 4, 4], num_res_blocks=2,
                 attention_resolutions=[4, 2, 1], num_head_channels=64, transformer_depth=1, context_dim=1024,
                 adm_in_channels=768, num_classes="sequential", use_linear_in_transformer=True, extra_ff_mix_layer=True,
                 use_spatial_context=True, merge_strategy="learned_with_images", video_kernel_size=[3, 1, 1],
                 use_checkpoint=False, spatial_transformer_attn_type="softmax")
FULL_CTRL = {k: v for k, v in FULL_UNET.items() if k != "out_channels"}
FULL_CTRL["hint_channels"] = 7
FULL_T, FULL_HW = 14, (72, 128)
FULL_PROBES = ("input_blocks.1", "input_blocks.7", "middle_block", "output_blocks.10")     # recorded subsampled [::7, ::16, ::4, ::4]
FULL_SUB = (slice(None, None, 7), slice(None, None, 16), slice(None, None, 4), slice(None, None, 4))
FULL_SAMPLE_STEPS = 2          # steps of the sampling loop recorded at full size (tools/gen_golden_sgm_full_sample.py)


def seeded_state_dict(module, seed):
    """Every parameter/buffer re-drawn (zero-initialised ones too, SURVEY.md §8c caveat) in sorted-key order."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k in sorted(module.state_dict().keys()):
        v = module.state_dict()[k]
        if not v.dtype.is_floating_point:
            sd[k] = v.clone()
            continue
        r = torch.randn(v.shape, generator=g)
        if k.endswith("mix_factor"):
            sd[k] = r
        elif v.ndim >= 2:
            fan_in = v[0].numel()
            sd[k] = r / fan_in ** 0.5
        elif k.endswith("weight"):
            sd[k] = 1.0 + 0.1 * r                       # norm scales
        else:
            sd[k] = 0.1 * r                             # biases
    return sd


def seeded_inputs(seed, T=T_FRAMES, hw=LATENT_HW, cfg=SMALL_UNET, cfg_doubled=True):
    g = torch.Generator().manual_seed(seed)
    B = (2 if cfg_doubled else 1) * T
    h, w = hw
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(
        x=r(B, 4, h, w), concat=r(B, 4, h, w), crossattn=r(B, 1, cfg["context_dim"]), vector=r(B, cfg["adm_in_channels"]),
        control_hint=torch.rand(B, 7, 8 * h, 8 * w, generator=g), sigma=torch.exp(r(B) * 1.2),
        image_only_indicator=torch.zeros(1, T), num_video_frames=T)


# ---- first-stage autoencoder: small config of configs/test/svd_f_est_ctrl_simp1.yaml:131-159
SMALL_VAE = dict(attn_type="vanilla", double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32,
                 ch_mult=[1, 2, 4], num_res_blocks=1, attn_resolutions=[], dropout=0.0)
VAE_T = 3
VAE_HW = (32, 16)
VAE_SAMPLE_SEED = 5


def vae_inputs(seed, T=VAE_T, hw=VAE_HW):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(T, 3, *hw, generator=g) * 2 - 1


# FULL-SIZE first stage (configs/test/svd_f_est_ctrl_simp1.yaml:131-159: ch 128, ch_mult [1, 2, 4, 4], 2 ResBlocks per level) at the
# 576x1024 frame size of configs[3]: the encoder on one frame, the video decoder on FULL_VAE_T latent frames of 72x128 —
# tests/golden/vae_full.npz (tools/gen_golden_vae_full.py). Outputs recorded subsampled FULL_VAE_SUB plus one dense 32x32 crop.
FULL_VAE = dict(attn_type="vanilla", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
FULL_VAE_T = 2
FULL_VAE_HW = (576, 1024)
FULL_VAE_SUB = (slice(None), slice(None), slice(None, None, 4), slice(None, None, 4))
FULL_VAE_CROP = (slice(None), slice(None), slice(272, 304), slice(496, 528))


def vae_full_latent(seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(FULL_VAE_T, 4, FULL_VAE_HW[0] // 8, FULL_VAE_HW[1] // 8, generator=g)
