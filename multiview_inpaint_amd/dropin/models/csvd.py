"""svd_inpaint1/models/csvd.py network classes (YAML targets `models.csvd.ControlledVideoUNet`,
`models.csvd.ControlNet`, configs/test/svd_f_est_ctrl_simp1.yaml:20,42) + the Lightning-free engine."""
from multiview_inpaint_amd.svd.unet import ControlledVideoUNet, ControlNet  # noqa: F401
from multiview_inpaint_amd.svd.engine import SVDInpaintEngine  # noqa: F401
