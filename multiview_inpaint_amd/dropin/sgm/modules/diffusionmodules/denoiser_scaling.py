from multiview_inpaint_amd.svd.schedule import (DenoiserScaling, EDMScaling, EpsScaling, VScaling,  # noqa: F401
                                                VScalingWithEDMcNoise)
