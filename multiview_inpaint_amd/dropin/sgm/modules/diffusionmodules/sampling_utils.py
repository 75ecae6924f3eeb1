from multiview_inpaint_amd.svd.schedule import to_d  # noqa: F401
