"""Namespace-stub recipe of SURVEY.md Appendix A: makes the reference's denoise-loop modules importable in the BUILD
container (never on the GPU box: /root/reference does not travel). Shared by the golden generators."""
import importlib
import sys
import types

import torch.nn as nn

ROOT = "/root/reference/svd_inpaint1"


def _ns(name, path=None, **attrs):
    m = types.ModuleType(name)
    m.__package__ = name
    if path:
        m.__path__ = [path]
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns a dict of the reference classes the generators use."""
    _ns("sgm", ROOT + "/sgm")
    _ns("sgm.modules", ROOT + "/sgm/modules", UNCONDITIONAL_CONFIG={})
    _ns("sgm.modules.diffusionmodules", ROOT + "/sgm/modules/diffusionmodules")
    _ns("omegaconf", ListConfig=list, OmegaConf=dict)
    _ns("torchvision")
    _ns("pytorch_lightning", LightningModule=nn.Module)
    _ns("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
    _ns("sgm.models")
    _ns("sgm.models.diffusion", DiffusionEngine=type("DiffusionEngine", (nn.Module,), {}))
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        _ns("matplotlib", pyplot=types.ModuleType("pyplot"))
        _ns("matplotlib.pyplot")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    csvd = importlib.import_module("models.csvd")
    return dict(VideoUNet=VideoUNet, ControlNet=csvd.ControlNet, ControlledVideoUNet=csvd.ControlledVideoUNet)
