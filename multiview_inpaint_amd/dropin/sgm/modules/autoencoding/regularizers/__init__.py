"""sgm.modules.autoencoding.regularizers (yaml :130)."""
from multiview_inpaint_amd.svd.vae import DiagonalGaussianRegularizer  # noqa: F401
