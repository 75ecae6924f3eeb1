from multiview_inpaint_amd.svd.schedule import (BaseDiffusionSampler, EDMSampler, EDMSampler2, EDMSampler3,  # noqa: F401
                                                EulerEDMSampler, EulerEDMSampler2, EulerEDMSampler3, HeunEDMSampler,
                                                SingleStepDiffusionSampler)
