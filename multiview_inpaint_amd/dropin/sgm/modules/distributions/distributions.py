from multiview_inpaint_amd.svd.vae import DiagonalGaussianDistribution  # noqa: F401
