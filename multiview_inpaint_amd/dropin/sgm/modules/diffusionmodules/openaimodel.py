from multiview_inpaint_amd.svd.layers import (Downsample, ResBlock, Timestep, TimestepBlock,  # noqa: F401
                                              TimestepEmbedSequential, Upsample)
from multiview_inpaint_amd.svd.transformer import SpatialTransformer, SpatialVideoTransformer  # noqa: F401
