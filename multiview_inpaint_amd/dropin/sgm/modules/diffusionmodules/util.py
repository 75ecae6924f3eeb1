from multiview_inpaint_amd.svd.layers import (AlphaBlender, GroupNorm32, avg_pool_nd, conv_nd, linear,  # noqa: F401
                                              normalization, timestep_embedding, zero_module)
