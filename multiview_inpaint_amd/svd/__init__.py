"""SVD temporal-UNet denoise loop (path B of the hot path): EDM schedule + v-prediction denoiser +
per-frame linear guidance + Euler sampler, the VideoUNet / ControlNet networks, and the device ops
(fused GroupNorm+SiLU, attention) that run as hand-written HIP on MI355X.

Reference: svd_inpaint1/sgm/modules/diffusionmodules/*, svd_inpaint1/sgm/modules/{attention,
video_attention}.py, svd_inpaint1/models/csvd.py. The drop-in packages `sgm` and `models` under
multiview_inpaint_amd/dropin/ re-export these classes under the reference's dotted names."""
