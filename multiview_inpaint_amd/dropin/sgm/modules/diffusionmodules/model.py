"""sgm.modules.diffusionmodules.model (yaml :132): first-stage Encoder / Decoder and their blocks."""
from multiview_inpaint_amd.svd.vae import (AttnBlock, Decoder, Downsample, Encoder, Normalize, ResnetBlock,  # noqa: F401
                                           Upsample, make_attn, nonlinearity)
