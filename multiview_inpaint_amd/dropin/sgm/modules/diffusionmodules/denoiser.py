from multiview_inpaint_amd.svd.schedule import Denoiser, DiscreteDenoiser  # noqa: F401
