"""svd_inpaint1/sgm/modules/attention.py names -> multiview_inpaint_amd.svd.transformer"""
from multiview_inpaint_amd.svd.transformer import (BasicTransformerBlock, CrossAttention, FeedForward, GEGLU,  # noqa: F401
                                                   MemoryEfficientCrossAttention, Normalize, SpatialTransformer)
from multiview_inpaint_amd.svd.layers import zero_module  # noqa: F401

XFORMERS_IS_AVAILABLE = True   # "softmax-xformers" is served by the HIP attention kernel
SDP_IS_AVAILABLE = True
