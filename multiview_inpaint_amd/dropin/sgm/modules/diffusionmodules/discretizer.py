from multiview_inpaint_amd.svd.schedule import Discretization, EDMDiscretization  # noqa: F401
