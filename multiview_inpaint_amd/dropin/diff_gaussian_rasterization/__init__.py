"""Drop-in for the package the reference imports at gs-simp/gaussian_renderer/__init__.py:14.
Put `multiview_inpaint_amd/dropin` on PYTHONPATH and gs-simp's train.py,
inpaint_rec.py, gen_seq.py, render*.py run unchanged on MI355X."""
from multiview_inpaint_amd.raster import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
