"""Stand-in of gs-simp/arguments: the three argument groups the launcher instantiates on its parser and `extract`s
(arguments/__init__.py:47-96 in the reference) — only the fields multiview_inpaint_amd.train_views reads."""
from types import SimpleNamespace


class _Group:
    FIELDS = ()          # (name, default, short flag or None)

    def __init__(self, parser):
        g = parser.add_argument_group(type(self).__name__)
        for name, default, short in self.FIELDS:
            flags = ["--" + name] + (["-" + short] if short else [])
            if isinstance(default, bool):
                g.add_argument(*flags, action="store_true", default=default)
            else:
                g.add_argument(*flags, type=type(default), default=default)

    def extract(self, args):
        return SimpleNamespace(**{name: getattr(args, name) for name, _, _ in self.FIELDS})


class ModelParams(_Group):
    FIELDS = (("sh_degree", 2, None), ("source_path", "", "s"), ("model_path", "", "m"), ("white_background", False, "w"))


class OptimizationParams(_Group):
    FIELDS = (("iterations", 30, None), ("lambda_dssim", 0.2, None), ("densify_from_iter", 3, None), ("densify_until_iter", 20, None),
              ("densification_interval", 5, None), ("opacity_reset_interval", 3000, None), ("densify_grad_threshold", 2e-5, None),
              ("random_background", False, None), ("percent_dense", 0.01, None))


class PipelineParams(_Group):
    FIELDS = (("convert_SHs_python", False, None), ("compute_cov3D_python", False, None), ("debug", False, None))
