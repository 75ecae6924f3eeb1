"""svd_inpaint1/sgm/modules/video_attention.py names -> multiview_inpaint_amd.svd.transformer"""
from multiview_inpaint_amd.svd.transformer import SpatialVideoTransformer, VideoTransformerBlock  # noqa: F401
