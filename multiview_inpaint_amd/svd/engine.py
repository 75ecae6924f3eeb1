"""Network wrappers and the Lightning-free counterpart of the reference's sampling engine.

Reference: sgm/modules/diffusionmodules/wrappers.py:9-34 (IdentityWrapper / OpenAIWrapper),
models/csvd.py:1086-1152 (SVDEngine.apply_model), :1258-1277 (SVDEngine.sample), stock path
sgm/models/diffusion.py:313-328. The reference engines are pytorch-lightning modules that also
own the VAE, the conditioner and checkpoint plumbing (out of scope); this
harness owns exactly the per-step hot path: ControlNet -> scaled residuals -> ControlledVideoUNet,
wrapped by Denoiser and driven by the sampler.
"""
import contextlib
import os
from typing import Dict, Optional, Sequence

import torch
import torch.nn as nn

from .schedule import Denoiser, instantiate_from_config
from .layers import to_planes

OPENAIUNETWRAPPER = "sgm.modules.diffusionmodules.wrappers.OpenAIWrapper"
# ControlNet beside the UNet encoder (apply_model): same-box A/B 181.3 -> 173.2 ms per step in round 3, 137.3 -> 133.7 in the driver's
# round-5 run. What stood in the way of making it the default (tools/experiments/two_stream_probe.py, every run under `timeout`): with
# hipBLASLt choosing its GEMM kernels by its own heuristic, one of the first steps of a process stops making progress (twice out of
# two runs; some of its stream-K kernels spin on flags of peer workgroups, and two of them on concurrent streams can keep each other's
# peers off the chip); with the GEMM set pinned by the shipped TunableOp file (svd/tunableop_gfx950.csv) the same sequence and every
# bench run since round 3 completed. So since round 6 the mode is GATED ON THAT CONDITION instead of on an environment variable:
#   MVI_SVD_TWO_STREAMS unset  -> on exactly when gemm_set_pinned(): TunableOp is enabled in look-up-only mode in this process
#                                 (bench_svd.enable_gemm_tuning(), what bench.py and tools/ do) and the file's validator lines equal
#                                 the running stack's (otherwise PyTorch ignores the file and the heuristic picks again);
#   MVI_SVD_TWO_STREAMS=1 / 0  -> forced on / off (A/B runs; forcing it on without the pinned set is the caller's risk).
# A process that never pinned its GEMMs (a plain `import` of the engine) therefore runs one stream, as before.
_FORCED = {"1": True, "0": False}.get(os.environ.get("MVI_SVD_TWO_STREAMS", ""))
TWO_STREAMS = _FORCED          # None: decided by gemm_set_pinned() at the first GPU step; tests and bench_svd assign True / False
_pinned = None


def gemm_set_pinned() -> bool:
    """True when every library GEMM of this process is looked up in the shipped TunableOp selection (no run-time tuning, validators of
    the file == validators of the running PyTorch / hipBLASLt / rocBLAS stack). Cached after the first call on a GPU."""
    global _pinned
    if _pinned is None:
        _pinned = False
        try:
            import torch.cuda.tunable as tn
            if tn.is_enabled() and not tn.tuning_is_enabled():
                want = {}
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")) as fh:
                    for line in fh:
                        if line.startswith("Validator,"):
                            _, k, v = line.rstrip("\n").split(",", 2)
                            want[k] = v
                have = {str(k): str(v) for k, v in (tn.get_validators() or ())}
                _pinned = bool(want) and bool(have) and all(have.get(k) == v for k, v in want.items())
        except Exception:
            _pinned = False
    return _pinned


def two_streams_active() -> bool:
    """Whether apply_model runs the ControlNet on a side stream in this process right now."""
    return gemm_set_pinned() if TWO_STREAMS is None else bool(TWO_STREAMS)


_side = {}


def _events_off():
    """The per-op timing of hip_ops (two timing events per op, hip_ops.PROFILE) and a second stream do not go together: with
    both on, the second step of tools/experiments/two_stream_probe.py never completed. Timed passes run on one stream."""
    from . import hip_ops
    return hip_ops.PROFILE is None


def _side_stream(device):
    s = _side.get(device.index)
    if s is None:
        s = _side[device.index] = torch.cuda.Stream(device)
    return s


class IdentityWrapper(nn.Module):
    def __init__(self, diffusion_model, compile_model: bool = False):
        super().__init__()
        if compile_model:
            raise NotImplementedError("no tracing compiler on this path: hot ops are hand-written HIP kernels")
        self.diffusion_model = diffusion_model

    def forward(self, *args, **kwargs):
        return self.diffusion_model(*args, **kwargs)


class OpenAIWrapper(IdentityWrapper):
    """x || c['concat'] on channels; crossattn -> context, vector -> y (wrappers.py:23-34)."""

    def forward(self, x, t, c: dict, **kwargs):
        cc = c.get("concat")
        if cc is not None and cc.numel():
            x = torch.cat((x, cc.type_as(x)), dim=1)
        return self.diffusion_model(x, timesteps=t, context=c.get("crossattn"), y=c.get("vector"), **kwargs)


class SVDInpaintEngine(nn.Module):
    """`model.diffusion_model` (ControlledVideoUNet) + `control_model` (ControlNet) + denoiser + sampler.
    Attribute names follow the reference engine so its checkpoints' key prefixes line up."""

    def __init__(self, network: nn.Module, control_model: Optional[nn.Module], denoiser: Denoiser, sampler=None,
                 control_scales: Optional[Sequence[float]] = None, global_average_pooling: bool = False):
        super().__init__()
        self.model = OpenAIWrapper(network)
        self.control_model = control_model
        self.denoiser = denoiser
        self.sampler = sampler
        self.control_scales = list(control_scales) if control_scales is not None else [1.0] * 13
        self.global_average_pooling = global_average_pooling

    @classmethod
    def from_configs(cls, network_config, control_config, denoiser_config, sampler_config=None, **kw):
        return cls(instantiate_from_config(network_config),
                   instantiate_from_config(control_config) if control_config else None,
                   instantiate_from_config(denoiser_config),
                   instantiate_from_config(sampler_config) if sampler_config else None, **kw)

    @property
    def device(self):
        return next(self.parameters()).device

    def apply_model(self, x, timesteps, cond: Dict, time_context=None, num_video_frames=None, image_only_indicator=None):
        cc = cond.get("concat")
        if cc is None:
            cc = x.new_zeros(x.shape[0], 0, *x.shape[2:])
        if "concat_scale" in cond:
            cc = cc * cond["concat_scale"]
        xin = torch.cat([x, cc.type_as(x)], dim=1)
        context = cond.get("crossattn")
        if "crossattn_scale" in cond:
            context = context * cond["crossattn_scale"]
        y = cond.get("vector")
        hint = cond.get("control_hint")
        if not torch.is_autocast_enabled():
            # parameters stored in reduced precision (no autocast): the sampler state stays fp32, the
            # network sees its own dtype; timesteps stay fp32 for the sinusoidal embedding
            wd = self.model.diffusion_model.time_embed[0].weight.dtype
            cast = lambda t: t.to(wd) if torch.is_tensor(t) and t.is_floating_point() else t
            xin, context, y = cast(xin), cast(context), cast(y)
            hint = [cast(h) for h in hint] if isinstance(hint, list) else cast(hint)
        if "palette" in cond:
            hint = [hint, cond["palette"]]
        controls = None
        if hint is not None and self.control_model is not None:
            def run_control():
                # (our ControlNet hands its residuals over token-major where it computed them so — layers.Tok — and the UNet consumes them so)
                extra = dict(tokens_out=True) if getattr(self.control_model, "offers_token_residuals", False) else {}
                cs = self.control_model(x=xin, hint=hint, timesteps=timesteps, context=context, y=y,
                                        time_context=time_context, num_video_frames=num_video_frames,
                                        image_only_indicator=image_only_indicator, **extra)
                cs = [c if s == 1.0 else c * s for c, s in zip(cs, self.control_scales)]   # x * 1.0 is x: skip the pass
                if self.global_average_pooling:
                    cs = [to_planes(c).mean(dim=(2, 3), keepdim=True) for c in cs]
                return cs
            if xin.is_cuda and not torch.is_grad_enabled() and _events_off() and two_streams_active():
                # The ControlNet and the UNet's encoder + middle block are independent until the first residual is added
                # (csvd.py:79): the ControlNet runs on a side stream while the main stream runs the encoder, so that the
                # low-resolution halves of both — whose kernels fill a quarter of the chip each (80 tiles per 3x3 convolution
                # at 9 x 16) — share it. The UNet joins the side stream where it pops the first residual.
                main = torch.cuda.current_stream(xin.device)
                side = _side_stream(xin.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    produced = run_control()
                for t in (xin, context, y, timesteps, *(hint if isinstance(hint, list) else [hint])):
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(side)

                def controls():
                    main.wait_stream(side)
                    for c in produced:
                        c.record_stream(main)
                    return produced
            else:
                controls = run_control()
        return self.model.diffusion_model(x=xin, timesteps=timesteps, context=context, y=y, time_context=time_context,
                                          control=controls, num_video_frames=num_video_frames,
                                          image_only_indicator=image_only_indicator)

    def denoise(self, x, sigma, cond, **kwargs):
        """One denoise step = Denoiser.forward over apply_model (csvd.py:1271-1273)."""
        return self.denoiser(self.apply_model, x, sigma, cond, **kwargs)

    @torch.no_grad()
    def sample(self, x, cond: Dict, uc: Optional[Dict] = None, batch_size: int = 16, shape=None, **kwargs):
        randn = torch.randn(batch_size, *shape).to(self.device)     # global torch RNG, as csvd.py:1269
        fn = lambda inp, sigma, c: self.denoise(inp, sigma, c, **kwargs)
        cache = getattr(self.control_model, "hint_cache", contextlib.nullcontext)
        if not torch.is_autocast_enabled():
            # reduced-precision parameters without autocast: cast the step-invariant conditioning ONCE per sample, not
            # in every apply_model call (a fresh 0.7 GB hint tensor per step would also miss the hint-stem cache, which
            # is keyed on the hint's storage)
            wd = self.model.diffusion_model.time_embed[0].weight.dtype
            once = lambda d: None if d is None else {k: (v.to(wd) if torch.is_tensor(v) and v.is_floating_point() and k in
                                                         ("control_hint", "concat", "crossattn", "vector") else v) for k, v in d.items()}
            shared_hint = uc is not None and cond.get("control_hint") is not None and uc.get("control_hint") is cond.get("control_hint")
            cond, uc = once(cond), once(uc)
            if shared_hint:
                uc["control_hint"] = cond["control_hint"]           # still ONE tensor in both halves, as the caller passed it
        with cache():                                               # the hint stem runs once per sample, not per step
            out = self.sampler(fn, randn, cond, uc=uc)              # (the sampler drops the guider's doubled conditioning)
        if out.is_cuda:
            from . import hip_ops
            hip_ops.check_groupnorm_cluster(out.device)             # one 4-byte read per sample: raises if a GroupNorm block
        return out                                                  # ever gave up waiting for its group (never silently wrong)
