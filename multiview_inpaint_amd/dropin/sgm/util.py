"""sgm.util names used on the denoise path (svd_inpaint1/sgm/util.py:137-199)."""
from multiview_inpaint_amd.svd.schedule import (append_dims, append_zero, default, get_obj_from_str,  # noqa: F401
                                                instantiate_from_config)


def exists(x):
    return x is not None


def count_params(model, verbose=False):
    n = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {n * 1.e-6:.2f} M params.")
    return n
