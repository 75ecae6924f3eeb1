"""sgm.models.autoencoder.AutoencodingEngine (svd_inpaint1/configs/test/svd_f_est_ctrl_simp1.yaml:125): encode / decode."""
from multiview_inpaint_amd.svd.vae import AutoencodingEngine  # noqa: F401
