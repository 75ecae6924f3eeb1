"""sgm.modules.autoencoding.temporal_ae (yaml :146): VideoDecoder and its temporal blocks."""
from multiview_inpaint_amd.svd.vae import (AE3DConv, MemoryEfficientVideoBlock, VideoBlock, VideoDecoder,  # noqa: F401
                                           VideoResBlock, make_time_attn)
