"""Stand-in of gs-simp/gaussian_renderer: the name dropin.patch_gs_simp wraps (the view-sharded trainer renders through
multiview_inpaint_amd.raster on the stored parameters and never calls this body)."""


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    raise NotImplementedError("stand-in render: not reached by the launcher test")
