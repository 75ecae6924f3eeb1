"""`simple_knn._C` surface: distCUDA2 (the only symbol the reference uses)."""
from multiview_inpaint_amd.train_ops import distCUDA2  # noqa: F401
