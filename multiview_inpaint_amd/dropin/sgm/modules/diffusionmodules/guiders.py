from multiview_inpaint_amd.svd.schedule import (Guider, IdentityGuider, LinearPredictionGuider,  # noqa: F401
                                                LinearPredictionGuider2, VanillaCFG)
