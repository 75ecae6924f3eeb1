from multiview_inpaint_amd.svd.layers import VideoResBlock  # noqa: F401
from multiview_inpaint_amd.svd.unet import VideoUNet  # noqa: F401
