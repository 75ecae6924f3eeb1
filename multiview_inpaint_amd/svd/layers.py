"""Convolutional building blocks of the video UNet: GroupNorm32(+SiLU), sinusoidal embeddings,
alpha blending of spatial/temporal branches, up/down-sampling, the spatial ResBlock and its
spatial+temporal VideoResBlock.

Reference: sgm/modules/diffusionmodules/util.py:207-231 (timestep_embedding), :259-276
(normalization/GroupNorm32), :312-372 (AlphaBlender); openaimodel.py:72-104
(TimestepEmbedSequential), :107-207 (Upsample/Downsample), :210-354 (ResBlock);
video_model.py:12-81 (VideoResBlock). Parameter names (state-dict keys) are identical to the
reference so svd.safetensors / ControlNet checkpoints load (sgm/models/diffusion.py:105).
"""
import math
import os
from dataclasses import dataclass
from typing import Iterable, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint as _torch_checkpoint

from . import ops


def zero_module(m: nn.Module) -> nn.Module:
    for p in m.parameters():
        p.detach().zero_()
    return m


def conv_nd(dims, *a, **k):
    return {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}[dims](*a, **k)


def avg_pool_nd(dims, *a, **k):
    return {1: nn.AvgPool1d, 2: nn.AvgPool2d, 3: nn.AvgPool3d}[dims](*a, **k)


def linear(*a, **k):
    return nn.Linear(*a, **k)


def maybe_checkpoint(fn, enabled, *args):
    """Activation checkpointing only matters when a graph is being recorded."""
    if enabled and torch.is_grad_enabled():
        return _torch_checkpoint(fn, *args, use_reentrant=False)
    return fn(*args)


_FREQS = {}


def _embedding_freqs(half, max_period, device):
    """The frequency table, computed on the CPU as the reference does (same values to the last bit) but uploaded once per
    device: the reference's per-call pageable upload is a host synchronisation in every denoise step and cannot be
    captured into a HIP graph."""
    key = (half, max_period, str(device))
    f = _FREQS.get(key)
    if f is None:
        f = _FREQS[key] = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half).to(device)
    return f


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """[N] -> [N, dim], cosine half first, then sine (util.py:207-231)."""
    if repeat_only:
        return timesteps[:, None].expand(-1, dim)
    half = dim // 2
    freqs = _embedding_freqs(half, max_period, timesteps.device)
    ang = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class GroupNorm32(nn.GroupNorm):
    """GroupNorm computed in fp32 and cast back (util.py:274-276); HIP kernel on the GPU."""

    def forward(self, x, silu: bool = False, chan_bias=None):
        return ops.group_norm(x, self.num_groups, self.weight, self.bias, self.eps, silu=silu, chan_bias=chan_bias)

    def forward_tokens(self, x, silu: bool = False, chan_bias=None):
        """The same norm with token-major output [N, (h w), C]."""
        return ops.group_norm_tokens(x, self.num_groups, self.weight, self.bias, self.eps, silu=silu, chan_bias=chan_bias)


def normalization(channels):
    return GroupNorm32(32, channels)


def conv_no_bias(conv, x, bias=None):
    """The convolution with its bias withheld (or replaced): the caller folds the bias into the next fused op,
    because PyTorch-ROCm otherwise adds it in a separate broadcast pass over the whole activation."""
    return conv._conv_forward(x, conv.weight, bias)


def norm_act(seq: nn.Sequential, x):
    """Runs a `[GroupNorm32, SiLU, ...]` prefix as ONE fused op; returns the activated tensor."""
    return seq[0](x, silu=True)


TOKEN_STREAM = os.environ.get("MVI_SVD_TOKEN_STREAM", "1") != "0"


class Tok:
    """The residual stream BETWEEN the blocks of the video UNet / ControlNet in token-major form (round 6, MVI_SVD_TOKEN_STREAM): the
    activation `b c h w` of the reference held as t [N, H W, C], plus — when the kernel that produced it left them — the GroupNorm
    statistics of t (hip_ops.GnPartials) for the norm that opens the next block. Every block of the SVD networks works on tokens
    inside (implicit-GEMM convolutions, attention), so carrying tokens across the block boundary removes the two layout passes per
    block and lets the block tails (skip add, AlphaBlender, `x + x_in`, the decoder's concatenation) be plain row kernels that
    also take the next norm's statistics (csrc/groupnorm_tokens.hip gt_fused_kernel). `shape` is the LOGICAL b c h w shape, so
    shape checks written for planes read the same."""
    __slots__ = ("t", "H", "W", "stats")
    takes_tokens = True

    def __init__(self, t, H, W, stats=None):
        self.t, self.H, self.W, self.stats = t, int(H), int(W), stats

    @property
    def shape(self):
        return torch.Size((self.t.shape[0], self.t.shape[2], self.H, self.W))

    dtype = property(lambda self: self.t.dtype)
    device = property(lambda self: self.t.device)
    is_cuda = property(lambda self: self.t.is_cuda)

    def dim(self):
        return 4

    def gn_stats(self, groups):
        """The producer's statistics if they are those of GroupNorm(groups) of t without a channel bias, else None."""
        st = self.stats
        return st if GN_STATS_FROM_TAILS and st is not None and st.groups == int(groups) and st.chan_bias is None else None

    def planes(self):
        """b c h w (one layout pass)."""
        N, S, C = self.t.shape
        if C % 8 == 0 and S % 8 == 0:
            from . import hip_ops
            return hip_ops.tokens_to_planes_add(self.t, None, spatial=(self.H, self.W))
        return self.t.transpose(1, 2).reshape(N, C, self.H, self.W).contiguous()

    def __mul__(self, s):
        return Tok(self.t * s, self.H, self.W)

    def record_stream(self, stream):
        self.t.record_stream(stream)


GN_STATS_FROM_TAILS = os.environ.get("MVI_SVD_GN_STATS_FROM_TAILS", "1") != "0"     # 0: the block tails leave no statistics (A/B)


def token_stream_ok(x):
    """Whether a forward pass on x may carry its residual stream token-major: the reduced-precision inference path on the GPU with the
    implicit-GEMM convolutions on (every block then works on tokens inside)."""
    return (TOKEN_STREAM and CONV_N320 and NHWC_CONVS and TIME_STACK_TOKENS and torch.is_tensor(x) and x.is_cuda and x.dim() == 4
            and x.dtype in (torch.bfloat16, torch.float16) and not torch.is_grad_enabled())


def to_tok(x):
    """b c h w -> Tok (one layout pass); a Tok is returned as it is."""
    if isinstance(x, Tok):
        return x
    from . import hip_ops
    N, C, H, W = x.shape
    if C % 8 == 0 and (H * W) % 8 == 0:
        return Tok(hip_ops.planes_to_tokens(x), H, W)
    return Tok(x.flatten(2).transpose(1, 2).contiguous(), H, W)


def to_planes(x):
    return x.planes() if isinstance(x, Tok) else x


def _conv1x1_as_rows(conv):
    return (isinstance(conv, nn.Conv2d) and tuple(conv.kernel_size) == (1, 1) and tuple(conv.stride) == (1, 1)
            and tuple(conv.padding) == (0, 0) and conv.groups == 1)


class Timestep(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, t):
        return timestep_embedding(t, self.dim)


class AlphaBlender(nn.Module):
    """alpha * spatial + (1 - alpha) * temporal with alpha = sigmoid(mix_factor), forced to 1 for
    image-only frames (util.py:312-372), including the CFG patch that repeats alpha when the batch
    was doubled after image_only_indicator was built (:365-367)."""
    strategies = ["learned", "fixed", "learned_with_images"]

    def __init__(self, alpha: float, merge_strategy: str = "learned_with_images",
                 rearrange_pattern: str = "b t -> (b t) 1 1"):
        super().__init__()
        assert merge_strategy in self.strategies, f"merge_strategy needs to be in {self.strategies}"
        self.merge_strategy, self.rearrange_pattern = merge_strategy, rearrange_pattern
        if merge_strategy == "fixed":
            self.register_buffer("mix_factor", torch.tensor([float(alpha)]))
        else:
            self.register_parameter("mix_factor", nn.Parameter(torch.tensor([float(alpha)])))

    def get_alpha(self, image_only_indicator, rows=None):
        """`rows`: the batch the blend is applied to (videos, i.e. leading size of the arranged alpha). When the indicator was built
        before the batch was doubled for classifier-free guidance the reference repeats alpha (util.py:365-367); with `rows` given
        that repeat happens here, once per indicator, instead of as a small concatenation in every block of every step."""
        if self.merge_strategy == "fixed":
            return self.mix_factor
        if self.merge_strategy == "learned":
            return torch.sigmoid(self.mix_factor)
        assert image_only_indicator is not None, "need image_only_indicator ..."
        if rows is not None:
            a = self.get_alpha(image_only_indicator)
            if a.size(0) == rows or torch.is_grad_enabled():
                return a if a.size(0) == rows else torch.cat([a] * 2)
            hit = self.__dict__.get("_alpha2_cache")
            if hit is not None and hit[0] is a:
                return hit[1]
            a2 = torch.cat([a] * 2)
            self.__dict__["_alpha2_cache"] = (a, a2)                 # keyed on the cached single-batch alpha object itself
            return a2
        # five tiny kernels per call and ~55 calls per denoise step: without a graph being recorded the result is kept until
        # the indicator or the mix factor changes (storage address + in-place version counters; the cache holds the
        # indicator it was computed from, so that address cannot be handed to another tensor meanwhile)
        m, cacheable = self.mix_factor, not torch.is_grad_enabled()
        key = (image_only_indicator.data_ptr(), image_only_indicator._version, tuple(image_only_indicator.shape),
               image_only_indicator.dtype, m.data_ptr(), m._version, m.dtype)
        hit = self.__dict__.get("_alpha_cache")
        if cacheable and hit is not None and hit[0] == key:
            return hit[1]
        ind = image_only_indicator.bool()
        a = torch.where(ind, torch.ones(1, 1, device=ind.device), torch.sigmoid(self.mix_factor)[..., None])
        a = self._arrange(a)
        if cacheable:
            self.__dict__["_alpha_cache"] = (key, a, image_only_indicator)
        return a

    def _arrange(self, a):
        b, t = a.shape
        if self.rearrange_pattern == "b t -> (b t) 1 1":
            return a.reshape(b * t, 1, 1)
        if self.rearrange_pattern == "b t -> b 1 t 1 1":
            return a.reshape(b, 1, t, 1, 1)
        from einops import rearrange
        return rearrange(a, self.rearrange_pattern)

    def forward(self, x_spatial, x_temporal, image_only_indicator=None):
        a = self.get_alpha(image_only_indicator)
        if a.numel() > 1 and a.size(0) != x_spatial.size(0):
            a = self.get_alpha(image_only_indicator, rows=x_spatial.size(0))
        # alpha * spatial + (1 - alpha) * temporal as ONE pass: temporal + alpha * (spatial - temporal)
        return torch.lerp(x_temporal, x_spatial, a.to(x_spatial.dtype))


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1, third_up=False,
                 kernel_size=3, scale_factor=2):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.use_conv, self.dims, self.third_up, self.scale_factor = use_conv, dims, third_up, scale_factor
        if use_conv:
            self.conv = conv_nd(dims, self.channels, self.out_channels, kernel_size, padding=padding)

    takes_tokens = True

    def forward(self, x):
        assert x.shape[1] == self.channels
        s = self.scale_factor
        if isinstance(x, Tok):
            N, C, H, W = x.shape
            if self.dims != 3 and s == 2 and self.use_conv and _conv_n320_route_ok(self.conv, N, C, 2 * H, 2 * W, x.dtype):
                from . import hip_ops                   # (the upsampling is in the kernel's addressing: no 4x tensor)
                return Tok(hip_ops.conv3x3_n320(x.t, _tap_major_weight(self.conv.weight), self.conv.bias, H, W, up2=True), 2 * H, 2 * W)
            return to_tok(self.forward(x.planes()))
        if self.dims == 3:
            x = F.interpolate(x, ((s if self.third_up else 1) * x.shape[2], x.shape[3] * s, x.shape[4] * s), mode="nearest")
        else:
            if s == 2 and self.use_conv:
                y = conv3x3_planes_via_tokens(self.conv, x, upsample=2)      # upsampling folded into the layout change
                if y is not None:
                    return y
            x = F.interpolate(x, scale_factor=s, mode="nearest")
        return self.conv(x) if self.use_conv else x


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1, third_down=False):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.use_conv, self.dims = use_conv, dims
        stride = 2 if dims != 3 else ((2, 2, 2) if third_down else (1, 2, 2))
        if use_conv:
            self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=stride, padding=padding)
        else:
            assert self.channels == self.out_channels
            self.op = avg_pool_nd(dims, kernel_size=stride, stride=stride)

    takes_tokens = True

    def forward(self, x):
        assert x.shape[1] == self.channels
        if isinstance(x, Tok):
            N, C, H, W = x.shape
            if self.use_conv and _conv_n320_route_ok(self.op, N, C, H, W, x.dtype):
                from . import hip_ops
                Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
                return Tok(hip_ops.conv3x3_n320(x.t, _tap_major_weight(self.op.weight), self.op.bias, H, W, stride=2), Ho, Wo)
            return to_tok(self.forward(x.planes()))
        if self.use_conv:
            y = conv3x3_planes_via_tokens(self.op, x)                        # the stride-2 convolution on tokens
            if y is not None:
                return y
        return self.op(x)


@dataclass
class BlockArgs:
    """What a TimestepEmbedSequential hands to each member that wants more than `x`."""
    emb: Optional[torch.Tensor] = None
    context: Optional[torch.Tensor] = None
    image_only_indicator: Optional[torch.Tensor] = None
    time_context: Optional[torch.Tensor] = None
    num_video_frames: Optional[int] = None


class TimestepBlock(nn.Module):
    """Marker: forward(x, emb)."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Sequential whose members declare, through `takes`, which conditioning they consume
    (openaimodel.py:72-104 dispatches on isinstance instead; same call signature)."""

    @staticmethod
    def _member(layer, x, a):
        kind = getattr(layer, "takes", None)
        if kind == "video_res":
            return layer(x, a.emb, a.num_video_frames, a.image_only_indicator)
        if kind == "video_attn":
            return layer(x, a.context, a.time_context, a.num_video_frames, a.image_only_indicator)
        if kind == "attn":
            return layer(x, a.context)
        if kind == "emb" or isinstance(layer, TimestepBlock):
            return layer(x, a.emb)
        if isinstance(layer, nn.Conv2d):
            return conv_no_bias(layer, x, layer.bias)          # 1x1 convolutions of channels-last rows run as GEMMs
        return layer(x)

    def forward(self, x, emb=None, context=None, image_only_indicator=None, time_context=None, num_video_frames=None):
        a = BlockArgs(emb, context, image_only_indicator, time_context, num_video_frames)
        for layer in self:
            if isinstance(x, Tok) and not getattr(layer, "takes_tokens", False):
                # a member that knows planes only, inside a token-major stream: a 1x1 convolution (the ControlNet's zero convolutions,
                # csvd.py:252-256) is a GEMM on the token rows; anything else sees b c h w and its result is turned back
                if _conv1x1_as_rows(layer) and layer.weight.dtype == x.dtype:
                    x = Tok(ops.linear(x.t, layer.weight.reshape(layer.out_channels, layer.in_channels), layer.bias), x.H, x.W)
                else:
                    x = to_tok(self._member(layer, x.planes(), a))
            else:
                x = self._member(layer, x, a)
        return x


class ResBlock(TimestepBlock):
    """GN-SiLU-conv, + Linear(SiLU(emb)), GN-SiLU-(dropout)-conv, + skip (openaimodel.py:210-354).
    GN+SiLU pairs run as one fused op."""
    takes = "emb"

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, dims=2, use_checkpoint=False, up=False, down=False,
                 kernel_size=3, exchange_temb_dims=False, skip_t_emb=False):
        super().__init__()
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_conv, self.use_checkpoint = use_conv, use_checkpoint
        self.use_scale_shift_norm, self.exchange_temb_dims = use_scale_shift_norm, exchange_temb_dims
        pad = [k // 2 for k in kernel_size] if isinstance(kernel_size, Iterable) else kernel_size // 2

        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(),
                                       conv_nd(dims, channels, self.out_channels, kernel_size, padding=pad))
        self.updown = up or down
        if up:
            self.h_upd, self.x_upd = Upsample(channels, False, dims), Upsample(channels, False, dims)
        elif down:
            self.h_upd, self.x_upd = Downsample(channels, False, dims), Downsample(channels, False, dims)
        else:
            self.h_upd = self.x_upd = nn.Identity()

        self.skip_t_emb = skip_t_emb
        self.emb_out_channels = 2 * self.out_channels if use_scale_shift_norm else self.out_channels
        if skip_t_emb:
            assert not use_scale_shift_norm
            self.emb_layers, self.exchange_temb_dims = None, False
        else:
            self.emb_layers = nn.Sequential(nn.SiLU(), linear(emb_channels, self.emb_out_channels))
        self.out_layers = nn.Sequential(
            normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
            zero_module(conv_nd(dims, self.out_channels, self.out_channels, kernel_size, padding=pad)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, kernel_size, padding=pad)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    def forward(self, x, emb):
        return maybe_checkpoint(self._forward, self.use_checkpoint, x, emb)

    def _forward(self, x, emb):
        if not (self.updown or self.use_scale_shift_norm or self.skip_t_emb or self.exchange_temb_dims):
            return self._forward_fused(x, emb)
        h = norm_act(self.in_layers, x)
        if self.updown:
            h, x = self.h_upd(h), self.x_upd(x)
        h = self.in_layers[2](h)
        if self.skip_t_emb:
            e = torch.zeros_like(h)
        else:
            e = self.emb_layers(emb).type(h.dtype)
        e = e.reshape(e.shape + (1,) * (h.ndim - e.ndim))
        if self.use_scale_shift_norm:
            scale, shift = e.chunk(2, dim=1)
            h = self.out_layers[0](h) * (1 + scale) + shift
            h = self.out_layers[3](self.out_layers[2](F.silu(h)))
        elif self.exchange_temb_dims:
            h = norm_act(self.out_layers, h + e.transpose(1, 2))       # b t c ... -> b c t ...
            h = self.out_layers[3](self.out_layers[2](h))
        else:
            # h + emb is never materialised: the per-(sample, channel) bias is added inside the norm kernel
            h = self.out_layers[0](h, silu=True, chan_bias=e.reshape(e.shape[0], e.shape[1]))
            h = self.out_layers[3](self.out_layers[2](h))
        return self.skip_connection(x) + h


def temporal_conv3_stacked(x3, conv: nn.Conv3d, with_bias=True):
    """Conv3d with kernel (3,1,1), padding (1,0,0) along the frame axis, given its input already stacked
    as x3 [(b T), 3 Ci, H, W] = (frame t-1 | frame t | frame t+1) (ops.group_norm_frames(stack3=True)):
    y_t = W[..., 0] x_{t-1} + W[..., 1] x_t + W[..., 2] x_{t+1} is ONE 1x1 convolution with K = 3 Ci,
    instead of permuting to b c t h w and running an im2col 3-D convolution."""
    ci3 = x3.shape[1]
    w = conv.weight
    key = (w.data_ptr(), w._version, w.dtype, w.device)
    hit = getattr(conv, "_w_stacked", None)
    if hit is None or hit[0] != key:        # tap-major channels [Co, 3 Ci], rebuilt only when the weight changes
        hit = (key, w.detach()[:, :, :, 0, 0].permute(0, 2, 1).reshape(conv.out_channels, ci3).contiguous())
        conv._w_stacked = hit
    wt = hit[1] if not (torch.is_grad_enabled() and w.requires_grad) else w[:, :, :, 0, 0].permute(0, 2, 1).reshape(conv.out_channels, ci3)
    return F.conv2d(x3, wt.reshape(conv.out_channels, ci3, 1, 1), conv.bias if with_bias else None)


def _f32_param(p):
    """fp32 copy of a small parameter: cached per parameter on the GPU (hip_ops._f32), a plain cast elsewhere."""
    if p.is_cuda:
        from . import hip_ops
        return hip_ops._f32(p)
    return p.float()


_bias_sums = {}


def _sum_param(a, b):
    """a + b (fp32) for two small parameters, computed once per version of either (two bias vectors met by one fused add).
    Entries hold weak references to both parameters: ids and addresses recur after a model is freed."""
    import weakref
    key = (id(a), id(b))
    ver = (a.data_ptr(), a._version, b.data_ptr(), b._version, a.dtype, a.device)
    hit = _bias_sums.get(key)
    if hit is None or hit[0] != ver or hit[2]() is not a or hit[3]() is not b:
        hit = (ver, a.detach().float() + b.detach().float(), weakref.ref(a), weakref.ref(b))
        if len(_bias_sums) > 4096:
            _bias_sums.clear()
        _bias_sums[key] = hit
    return hit[1]


_silu_emb_cache = []          # [(weakref(emb), version, silu(emb))], newest first: UNet and ControlNet each have one embedding per step


def _emb_projection(emb_layers, emb):
    """emb_layers(emb) = Linear(SiLU(emb)) with SiLU(emb) computed once per embedding tensor instead of once per ResBlock
    (44 ResBlocks of a step share two embeddings); same values, same kernels."""
    import weakref
    if not (isinstance(emb_layers, nn.Sequential) and len(emb_layers) == 2 and isinstance(emb_layers[0], nn.SiLU)) \
            or emb.requires_grad or torch.is_grad_enabled() and any(p.requires_grad for p in emb_layers.parameters()):
        return emb_layers(emb)
    for ref, ver, act in _silu_emb_cache:
        if ref() is emb and ver == emb._version:
            return emb_layers[1](act)
    act = F.silu(emb)
    _silu_emb_cache.insert(0, (weakref.ref(emb), emb._version, act))
    del _silu_emb_cache[2:]
    return emb_layers[1](act)


BATCHED_EMB = os.environ.get("MVI_SVD_BATCHED_EMB", "1") != "0"
_emb_plans = {}               # id(root) -> (weakref(root), signature, groups)
_emb_tables = []              # [(weakref(emb), version, {id(linear): fp32 [N, C]})], newest first (UNet + ControlNet)


def _emb_plan(root):
    """Every (VideoRes)ResBlock under `root` that projects the step's embedding through Linear(SiLU(emb)) in its fused forward, grouped
    by output width: per group the concatenated weight / bias of the projections and the concatenated bias of the convolution
    each projection is added behind (conv1: it rides with the embedding inside the second GroupNorm). Cached per root and
    rebuilt when any of those parameters changes."""
    import weakref
    blocks = [m for m in root.modules() if isinstance(m, ResBlock) and isinstance(m.emb_layers, nn.Sequential) and len(m.emb_layers) == 2
              and isinstance(m.emb_layers[0], nn.SiLU) and type(m.emb_layers[1]) is nn.Linear and not m.use_scale_shift_norm
              and isinstance(m.in_layers[2], (nn.Conv2d, nn.Conv3d))]
    if len(blocks) < 4:
        return None
    params = []
    for b in blocks:
        lin, conv = b.emb_layers[1], b.in_layers[2]
        params += [lin.weight, lin.bias, conv.bias]
    if any(p is not None and p.requires_grad and torch.is_grad_enabled() for p in params):
        return None
    sig = tuple((id(p), p._version, p.data_ptr()) if p is not None else None for p in params)
    hit = _emb_plans.get(id(root))
    if hit is not None and hit[0]() is root and hit[1] == sig:
        return hit[2]
    by_width = {}
    for b in blocks:
        lin = b.emb_layers[1]
        if lin.bias is None or lin.out_features != b.in_layers[2].out_channels:
            continue
        by_width.setdefault((lin.out_features, lin.in_features, lin.weight.dtype, lin.weight.device), []).append(b)
    groups = []
    with torch.no_grad():
        for (C, _, _, dev), bs in by_width.items():
            W = torch.cat([b.emb_layers[1].weight for b in bs], 0).contiguous()
            bl = torch.cat([b.emb_layers[1].bias for b in bs], 0).contiguous()
            cb = torch.cat([(b.in_layers[2].bias.float() if b.in_layers[2].bias is not None else torch.zeros(C, device=dev)) for b in bs], 0)
            groups.append((C, W, bl, cb.contiguous(), [id(b.emb_layers[1]) for b in bs]))
    key = id(root)
    _emb_plans[key] = (weakref.ref(root, lambda _r, k=key: _emb_plans.pop(k, None)), sig, groups)
    return groups


def prepare_emb_projections(root, emb):
    """One GEMM per output width for ALL embedding projections of a network's ResBlocks (44 in the UNet, 25 in the ControlNet:
    each was a 28-row GEMM + a bias add of a few microseconds, ~130 launches per network and step), with the bias of the
    convolution in front folded in as the per-block path does (fp32 add behind the GEMM's rounding: the same values). The
    blocks pick their [N, C] fp32 slice up in _emb_chan_bias; anything not prepared here takes the per-block path."""
    import weakref
    if not (BATCHED_EMB and emb.is_cuda) or torch.is_grad_enabled() or emb.requires_grad or emb.dim() != 2:
        return
    groups = _emb_plan(root)
    if not groups:
        return
    act = F.silu(emb)
    N = emb.shape[0]
    table = {}
    for C, W, bl, cb, ids in groups:
        if W.dtype != act.dtype:
            return
        out = F.linear(act, W, bl).float() + cb                      # [N, n C] fp32
        out = out.view(N, len(ids), C).permute(1, 0, 2).contiguous()  # [n, N, C]: a block's slice is contiguous
        for k, i in enumerate(ids):
            table[i] = out[k]
    _emb_tables.insert(0, (weakref.ref(emb), emb._version, table))
    del _emb_tables[2:]


def _emb_chan_bias(emb_layers, emb, conv):
    """fp32 [N, C]: emb_layers(emb) + conv.bias — what the second GroupNorm of a ResBlock adds per (sample, channel). From the
    batched projection of this step when the network prepared one (prepare_emb_projections), else computed here."""
    if isinstance(emb_layers, nn.Sequential) and len(emb_layers) == 2:
        for ref, ver, table in _emb_tables:
            if ref() is emb and ver == emb._version:
                hit = table.get(id(emb_layers[1]))
                if hit is not None:
                    return hit
                break
    e = _emb_projection(emb_layers, emb)
    e = e.reshape(e.shape[0], e.shape[1])
    # fp32 [N, C] for the norm's chan_bias: the mixed-dtype add promotes inside one kernel (same values as cast-then-add)
    return e + _f32_param(conv.bias) if conv.bias is not None else e.float()


NHWC_CONVS = os.environ.get("MVI_SVD_NHWC_CONVS", "1") != "0"
_cl_weights = {}


def _channels_last_weight(w):
    """The convolution weight in channels-last memory format, converted once per parameter version."""
    key = id(w)
    hit = _cl_weights.get(key)
    if hit is None or hit[0]() is not w or hit[1] != (w.data_ptr(), w._version, w.dtype, w.device):
        import weakref
        hit = (weakref.ref(w, lambda _r, k=key: _cl_weights.pop(k, None)), (w.data_ptr(), w._version, w.dtype, w.device),
               w.detach().contiguous(memory_format=torch.channels_last))
        _cl_weights[key] = hit
    return hit[2]


CONV_N320 = os.environ.get("MVI_SVD_CONV_N320", "1") != "0"
CONV_N320_MIN_BLOCKS = 128         # fewer blocks of 256 rows x 320 channels than this leave most of the 256 CUs idle: the kernel splits K
                                   # then (level 3: 64 blocks x 4), or, for a K too short to split, the library runs
_tap_weights = {}


def _tap_major_weight(w):
    """A convolution weight in csrc/linear_n320.hip's implicit-GEMM order, [C_out][taps C_in] tap-major — 3x3 Conv2d weights
    ([C_out, C_in, 3, 3] -> 9 taps) and (3,1,1) Conv3d weights ([C_out, C_in, 3, 1, 1] -> 3 taps) — once per parameter version."""
    from . import hip_ops
    key = id(w)
    hit = _tap_weights.get(key)
    # the packing order of the 3x3 weights is a process-global switch of the kernel (mvi_conv3x3_n320_k_order, read again at every
    # launch): part of the version, so a switch after the first forward re-packs instead of silently mis-convolving
    ver = (w.data_ptr(), w._version, w.dtype, w.device, int(hip_ops._lib.lib().mvi_conv3x3_n320_k_order(-1)) if w.dim() == 4 else -1)
    if hit is None or hit[0]() is not w or hit[1] != ver:
        import weakref
        build = hip_ops.conv3t_n320_weight if w.dim() == 5 else hip_ops.conv3x3_n320_weight
        hit = (weakref.ref(w, lambda _r, k=key: _tap_weights.pop(k, None)), ver, build(w.detach()))
        _tap_weights[key] = hit
    return hit[2]


def conv3x3_planes_via_tokens(conv, x, upsample=1, tokens_out=False):
    """`conv(x)` — or `conv(F.interpolate(x, scale_factor=2, mode="nearest"))` with upsample = 2 — for a 3x3 / padding 1 Conv2d of
    stride 1 or 2 on an NCHW tensor, evaluated on tokens: layout pass (the upsampling folded into it), the implicit-GEMM kernel of
    csrc/linear_n320.hip with the bias in its accumulators, layout pass back. None when the route does not apply: reduced precision
    on the GPU outside autograd, a shape the kernel takes (C_out = 320 g, C_in = 64 k), enough output pixels to fill the chip (or a K
    long enough to split). Used by Upsample, Downsample and the last convolution of the ControlNet hint stem."""
    if not (CONV_N320 and NHWC_CONVS and x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float16)
            and not torch.is_grad_enabled() and isinstance(conv, nn.Conv2d) and conv.weight.dtype == x.dtype
            and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) in ((1, 1), (2, 2)) and tuple(conv.padding) == (1, 1)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1):
        return None
    from . import hip_ops
    N, C, H, W = x.shape
    stride = conv.stride[0]
    Hi, Wi = upsample * H, upsample * W
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    if not ((H * W) % 8 == 0 and (Ho * Wo) % 8 == 0 and _conv_n320_shape_ok(conv, N, C, Hi, Wi, x.dtype)):
        return None
    t = hip_ops.conv3x3_n320(hip_ops.planes_to_tokens(x, upsample=upsample), _tap_major_weight(conv.weight), conv.bias, Hi, Wi, stride=stride)
    if tokens_out:                                                       # (the caller carries the result token-major: layers.Tok)
        return Tok(t, Ho, Wo)
    return hip_ops.tokens_to_planes_add(t, None, spatial=(Ho, Wo))


def _conv_n320_shape_ok(conv, N, C, Hi, Wi, dtype):
    """A 3x3 / padding 1 convolution of stride 1 or 2 over N images of Hi x Wi tokens with C channels: the shapes csrc/linear_n320.hip takes
    and enough output pixels to fill the chip."""
    from . import hip_ops
    return (C % 8 == 0 and hip_ops.conv3x3_n320_supported(C, conv.out_channels, dtype) and N * Hi * Wi * C * 2 < 2 ** 32
            and hip_ops.conv3x3_n320_fills_chip(N, Hi, Wi, C, conv.out_channels, CONV_N320_MIN_BLOCKS, stride=conv.stride[0]))


def _conv_n320_route_ok(conv, N, C, Hi, Wi, dtype):
    """conv3x3_planes_via_tokens' conditions for a token-major input (Tok) of N x (Hi Wi) x C."""
    return (CONV_N320 and NHWC_CONVS and dtype in (torch.bfloat16, torch.float16) and not torch.is_grad_enabled()
            and isinstance(conv, nn.Conv2d) and conv.weight.dtype == dtype and tuple(conv.kernel_size) == (3, 3)
            and tuple(conv.stride) in ((1, 1), (2, 2)) and tuple(conv.padding) == (1, 1) and tuple(conv.dilation) == (1, 1)
            and conv.groups == 1 and _conv_n320_shape_ok(conv, N, C, Hi, Wi, dtype))


TIME_STACK_TOKENS = os.environ.get("MVI_SVD_TIME_STACK_TOKENS", "1") != "0"


GN_STATS_FROM_CONV = os.environ.get("MVI_SVD_GN_STATS_FROM_CONV", "1") != "0"      # 0: every token GroupNorm runs its own statistics pass (A/B)


def _conv_tokens(conv, tok, H, W, gn=None):
    """3x3 convolution of token-major activations [N, H W, C_in] -> [N, H W, C_out], bias withheld. Output channels in multiples
    of 320 (every ResBlock convolution of the SVD networks) go to the hand-written implicit GEMM; anything else is handed to the
    library as a channels-last view, so MIOpen's NHWC kernel runs without the transposes it wraps
    around NCHW tensors.
    gn = (groups, chan_bias): the GroupNorm that follows; returns (out, partials) — partials (hip_ops.GnPartials) when the
    implicit-GEMM launch could leave that norm's statistics behind (levels 0 and 1 of the 576 x 1024 step), else None."""
    N, S, C = tok.shape
    if CONV_N320:
        from . import hip_ops
        if (hip_ops.conv3x3_n320_supported(C, conv.out_channels, tok.dtype) and N * S * C * 2 < 2 ** 32
                and hip_ops.conv3x3_n320_fills_chip(N, H, W, C, conv.out_channels, CONV_N320_MIN_BLOCKS)):
            if gn is not None and GN_STATS_FROM_CONV and hip_ops.conv_n320_gnstats_supported(N * S, 9, C, conv.out_channels, S, gn[0]):
                return hip_ops.conv3x3_n320(tok, _tap_major_weight(conv.weight), None, H, W, gn=gn)
            y = hip_ops.conv3x3_n320(tok, _tap_major_weight(conv.weight), None, H, W)
            return (y, None) if gn is not None else y
    x = tok.view(N, H, W, C).permute(0, 3, 1, 2)                  # [N, C, H, W] with channels-last strides: no copy
    y = F.conv2d(x, _channels_last_weight(conv.weight), None, conv.stride, conv.padding, conv.dilation, conv.groups)
    if not y.is_contiguous(memory_format=torch.channels_last):     # (the library answered in NCHW: still correct, one copy)
        y = y.contiguous(memory_format=torch.channels_last)
    y = y.permute(0, 2, 3, 1).reshape(N, y.shape[2] * y.shape[3], y.shape[1])
    return (y, None) if gn is not None else y


def _nhwc_path_ok(self, x):
    conv1, conv2 = self.in_layers[2], self.out_layers[3]
    return (NHWC_CONVS and x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float16) and not torch.is_grad_enabled()
            and isinstance(conv1, nn.Conv2d) and isinstance(conv2, nn.Conv2d) and conv1.weight.dtype == x.dtype
            and all(tuple(c.kernel_size) == (3, 3) and tuple(c.stride) == (1, 1) and tuple(c.padding) == (1, 1) and c.groups == 1
                    for c in (conv1, conv2))
            and x.shape[1] % 8 == 0 and conv1.out_channels % 8 == 0 and conv2.out_channels % 8 == 0 and (x.shape[2] * x.shape[3]) % 8 == 0
            and _tok2tok_ok(x.shape[0], conv1.out_channels, x.shape[2] * x.shape[3], self.out_layers[0].num_groups, x.dtype))


def _tok2tok_ok(N, C, S, groups, dtype):
    """The token-major GroupNorm between the two convolutions has geometry limits of its own (rows per launch, groups, its
    row pass in LDS): a shape outside them keeps the whole block on the NCHW path instead of raising inside it."""
    from . import hip_ops
    return hip_ops._lib.lib().mvi_groupnorm_tok2tok_workspace_bytes(int(N), int(C), int(S), int(groups), hip_ops._DT[dtype]) != 0


def _planes_add_to_tokens(t, x, bias):
    from . import hip_ops
    return hip_ops.planes_add_to_tokens(x, t, bias)


def _resblock_forward_fused(self, x, emb, tokens_out=False):
    """The common ResBlock configuration (no up/down-sampling, additive embedding) with every bias and
    broadcast add folded into a neighbouring kernel: conv1's bias rides with the embedding bias inside the
    second GroupNorm, conv2's bias is added together with the skip tensor in one pass.
    In reduced precision on the GPU the two convolutions see channels-last tensors (MVI_SVD_NHWC_CONVS): the first norm writes
    tokens, the norm between the convolutions is token-major on both sides, and the last add reads tokens — the library's
    NHWC kernels then run without their NCHW <-> NHWC transposes (9 ms of a 14 x 576x1024 step)."""
    conv1, conv2 = self.in_layers[2], self.out_layers[3]
    if _nhwc_path_ok(self, x):
        g1, g2 = self.in_layers[0], self.out_layers[0]
        H, W = x.shape[2], x.shape[3]
        t = ops.group_norm_tokens(x, g1.num_groups, g1.weight, g1.bias, g1.eps, silu=True)
        e = _emb_chan_bias(self.emb_layers, emb, conv1)
        e = e if e.is_contiguous() else e.contiguous()
        t, stats = _conv_tokens(conv1, t, H, W, gn=(g2.num_groups, e))      # (the second norm's statistics ride in the convolution's epilogue)
        t = ops.group_norm_tok2tok(t, g2.num_groups, g2.weight, g2.bias, g2.eps, silu=True, chan_bias=e, partials=stats)
        t = _conv_tokens(conv2, self.out_layers[2](t), H, W)
        # tokens_out (VideoResBlock with a token-major temporal ResBlock behind): the same add with the result left token-major
        last_add = _planes_add_to_tokens if tokens_out else ops.tokens_to_planes_add
        if isinstance(self.skip_connection, nn.Identity):
            return last_add(t, x, conv2.bias)
        sk = self.skip_connection
        sb = sk.bias if conv2.bias is None else (conv2.bias if sk.bias is None else _sum_param(sk.bias, conv2.bias))
        return last_add(t, conv_no_bias(sk, x, None), sb)                      # both biases ride on the transposing add
    assert not tokens_out, "tokens_out is only offered on the channels-last route (_nhwc_path_ok)"
    h = conv_no_bias(conv1, norm_act(self.in_layers, x))
    e = _emb_chan_bias(self.emb_layers, emb, conv1)
    h = self.out_layers[0](h, silu=True, chan_bias=e)
    h = conv_no_bias(conv2, self.out_layers[2](h))
    if isinstance(self.skip_connection, nn.Identity):
        return ops.bias_residual_add(h, conv2.bias, x)
    sk = self.skip_connection
    sb = sk.bias if conv2.bias is None else (conv2.bias if sk.bias is None else sk.bias + conv2.bias)
    return h + conv_no_bias(sk, x, sb)


ResBlock._forward_fused = _resblock_forward_fused


class VideoResBlock(ResBlock):
    """Spatial ResBlock on (b t) c h w, then a temporal ResBlock (Conv3d kernel (3,1,1)) on
    b c t h w, blended by AlphaBlender (video_model.py:12-81)."""
    takes = "video_res"

    def __init__(self, channels, emb_channels, dropout, video_kernel_size=3, merge_strategy="fixed",
                 merge_factor=0.5, out_channels=None, use_conv=False, use_scale_shift_norm=False, dims=2,
                 use_checkpoint=False, up=False, down=False):
        super().__init__(channels, emb_channels, dropout, out_channels=out_channels, use_conv=use_conv,
                         use_scale_shift_norm=use_scale_shift_norm, dims=dims, use_checkpoint=use_checkpoint,
                         up=up, down=down)
        oc = out_channels if out_channels is not None else channels
        self.time_stack = ResBlock(oc, emb_channels, dropout=dropout, dims=3, out_channels=oc,
                                   use_scale_shift_norm=False, use_conv=False, up=False, down=False,
                                   kernel_size=video_kernel_size, use_checkpoint=use_checkpoint,
                                   exchange_temb_dims=True)
        self.time_mixer = AlphaBlender(alpha=merge_factor, merge_strategy=merge_strategy,
                                       rearrange_pattern="b t -> b 1 t 1 1")

    def _frames_path_ok(self):
        ts = self.time_stack
        conv = ts.in_layers[2]
        return (isinstance(conv, nn.Conv3d) and tuple(conv.kernel_size) == (3, 1, 1) and tuple(conv.padding) == (1, 0, 0)
                and tuple(conv.stride) == (1, 1, 1) and not ts.updown and not ts.use_scale_shift_norm and not ts.skip_t_emb
                and isinstance(ts.skip_connection, nn.Identity))

    def _time_stack_frames(self, x, emb, T, blend=None):
        """The temporal ResBlock (video_model.py:41-54, openaimodel.py:328-354 with dims=3 and
        exchange_temb_dims) evaluated on x [(b T), c, h, w]: temporal GroupNorm+SiLU with strided
        statistics, (3,1,1) convolutions as channel-stacked 1x1 convolutions, per-frame embedding bias."""
        ts = self.time_stack
        g0, g1 = ts.in_layers[0], ts.out_layers[0]
        c1, c2 = ts.in_layers[2], ts.out_layers[3]
        h3 = ops.group_norm_frames(x, T, g0.num_groups, g0.weight, g0.bias, g0.eps, silu=True, stack3=True)
        h = temporal_conv3_stacked(h3, c1, with_bias=False)
        e = _emb_chan_bias(ts.emb_layers, emb, c1)                 # [(b T), c] fp32: already per frame, fused into the norm
        h3 = ops.group_norm_frames(h, T, g1.num_groups, g1.weight, g1.bias, g1.eps, silu=True, chan_bias=e, stack3=True)
        h = temporal_conv3_stacked(ts.out_layers[2](h3), c2, with_bias=False)
        if blend is not None:                                      # AlphaBlender folded into the skip add
            return ops.bias_residual_blend(h, c2.bias, x, blend)
        return ops.bias_residual_add(h, c2.bias, x)

    def _tokens_path_ok(self, x, t):
        """The whole block on tokens (spatial ResBlock ending token-major, temporal ResBlock on tokens with the (3,1,1) convolutions
        in csrc/linear_n320.hip, b c h w restored by the blending tail): the channels-last route of the spatial block must apply, the
        temporal block must be the plain configuration, and the temporal convolutions shapes the implicit-GEMM kernel takes."""
        if not (TIME_STACK_TOKENS and CONV_N320 and self._frames_path_ok()           # (no autograd here: checkpointing is moot)
                and not (self.updown or self.use_scale_shift_norm or self.skip_t_emb or self.exchange_temb_dims) and _nhwc_path_ok(self, x)):
            return False
        from . import hip_ops
        ts = self.time_stack
        c1, c2 = ts.in_layers[2], ts.out_layers[3]
        bt, c, h, w = x.shape
        co = self.out_channels
        return (bt % t == 0 and c1.weight.dtype == x.dtype and c1.in_channels == co and c1.out_channels == co and c2.out_channels == co
                and hip_ops.conv3x3_n320_supported(co, co, x.dtype) and bt * h * w * co * 2 < 2 ** 32
                and _tok2tok_ok(bt, co, h * w, ts.in_layers[0].num_groups, x.dtype)
                and hip_ops.conv3t_n320_fills_chip(bt // t, t, h * w, co, co, CONV_N320_MIN_BLOCKS))

    def _time_stack_tokens(self, xt, emb, T, blend, hw, stats_in=None, tokens_out=False):
        """_time_stack_frames on token-major xt [(b T), S, c] (the spatial ResBlock's output as tokens): temporal GroupNorm + SiLU with
        token-major input and output, the (3,1,1) convolutions as three-tap implicit GEMMs over the frame axis, and the skip add +
        AlphaBlender in the pass that restores b c h w. No channel-stacked tensor, no library convolution."""
        from . import hip_ops
        ts = self.time_stack
        g0, g1 = ts.in_layers[0], ts.out_layers[0]
        c1, c2 = ts.in_layers[2], ts.out_layers[3]
        h = ops.group_norm_tok2tok(xt, g0.num_groups, g0.weight, g0.bias, g0.eps, silu=True, frames=T, partials=stats_in)
        e = _emb_chan_bias(ts.emb_layers, emb, c1)                 # [(b T), c] fp32 incl. the first convolution's bias
        e = e if e.is_contiguous() else e.contiguous()
        stats = None
        BT, S, C = h.shape
        if GN_STATS_FROM_CONV and hip_ops.conv_n320_gnstats_supported(BT * S, 3, C, c1.out_channels, S, g1.num_groups):
            h, stats = hip_ops.conv3t_n320(h, _tap_major_weight(c1.weight), None, T, gn=(g1.num_groups, e))
        else:
            h = hip_ops.conv3t_n320(h, _tap_major_weight(c1.weight), None, T)
        h = ops.group_norm_tok2tok(h, g1.num_groups, g1.weight, g1.bias, g1.eps, silu=True, chan_bias=e, frames=T, partials=stats)
        h = hip_ops.conv3t_n320(ts.out_layers[2](h), _tap_major_weight(c2.weight), None, T)
        if tokens_out:
            # the token-major stream: skip add + AlphaBlender as a row pass that leaves the statistics of the next block's first norm
            out, st = hip_ops.rows_fused(h, bias=c2.bias, base=xt, alpha=blend, groups=g0.num_groups if GN_STATS_FROM_TAILS else 0)
            return Tok(out, hw[0], hw[1], st)
        return hip_ops.tokens_blend_to_planes(h, xt, c2.bias, blend, hw)

    takes_tokens = True

    def _forward_tok(self, x, emb, t, image_only_indicator):
        """forward() on the token-major stream (Tok in, Tok out): first norm straight from tokens (its statistics, where the block
        before left them, cost no pass), the spatial block's skip add as a row pass that takes the temporal norm's statistics, the
        channel-changing skip convolution (1x1, openaimodel.py:300) as a GEMM on the token rows. None where a condition of the token
        path (_tokens_path_ok) fails: the caller goes through planes."""
        sk = self.skip_connection
        g1, g2 = self.in_layers[0], self.out_layers[0]
        N, C, H, W = x.shape
        if not (self._tokens_path_ok(x, t) and (isinstance(sk, nn.Identity) or (_conv1x1_as_rows(sk) and sk.weight.dtype == x.dtype))
                and _tok2tok_ok(N, C, H * W, g1.num_groups, x.dtype)):
            return None
        a = self.time_mixer.get_alpha(image_only_indicator)           # [b, 1, t, 1, 1] (or a scalar)
        if a.ndim != 5:
            return None
        if a.size(0) != N // t:
            a = self.time_mixer.get_alpha(image_only_indicator, rows=N // t)   # CFG-doubled batch (util.py:365-367)
        from . import hip_ops
        conv1, conv2 = self.in_layers[2], self.out_layers[3]
        h = ops.group_norm_tok2tok(x.t, g1.num_groups, g1.weight, g1.bias, g1.eps, silu=True, partials=x.gn_stats(g1.num_groups))
        e = _emb_chan_bias(self.emb_layers, emb, conv1)
        e = e if e.is_contiguous() else e.contiguous()
        h, stats = _conv_tokens(conv1, h, H, W, gn=(g2.num_groups, e))
        h = ops.group_norm_tok2tok(h, g2.num_groups, g2.weight, g2.bias, g2.eps, silu=True, chan_bias=e, partials=stats)
        h = _conv_tokens(conv2, self.out_layers[2](h), H, W)
        if isinstance(sk, nn.Identity):
            xs, sb = x.t, conv2.bias
        else:
            xs = ops.linear(x.t, sk.weight.reshape(sk.out_channels, sk.in_channels), None)
            sb = sk.bias if conv2.bias is None else (conv2.bias if sk.bias is None else _sum_param(sk.bias, conv2.bias))
        g0 = self.time_stack.in_layers[0]
        xt, st0 = hip_ops.rows_fused(h, xs, bias=sb, groups=g0.num_groups if GN_STATS_FROM_TAILS else 0)
        return self._time_stack_tokens(xt, emb, t, a.reshape(N), (H, W), stats_in=st0, tokens_out=True)

    def forward(self, x, emb, num_video_frames, image_only_indicator=None):
        t = int(num_video_frames)
        if isinstance(x, Tok):
            y = self._forward_tok(x, emb, t, image_only_indicator)
            return y if y is not None else to_tok(self.forward(x.planes(), emb, num_video_frames, image_only_indicator))
        if x.dim() == 4 and self._tokens_path_ok(x, t):
            a = self.time_mixer.get_alpha(image_only_indicator)           # [b, 1, t, 1, 1] (or a scalar)
            if a.ndim == 5:
                bt = x.shape[0]
                if a.size(0) != bt // t:
                    a = self.time_mixer.get_alpha(image_only_indicator, rows=bt // t)   # CFG-doubled batch (util.py:365-367)
                xt = self._forward_fused(x, emb, tokens_out=True)
                return self._time_stack_tokens(xt, emb, t, a.reshape(bt), tuple(x.shape[2:]))
        x = super().forward(x, emb)
        bt, c, h, w = x.shape
        if self._frames_path_ok():
            a = self.time_mixer.get_alpha(image_only_indicator)           # [b, 1, t, 1, 1] (or a scalar)
            if a.ndim == 5:
                if a.size(0) != bt // t:
                    a = self.time_mixer.get_alpha(image_only_indicator, rows=bt // t)   # CFG-doubled batch (util.py:365-367)
                return self._time_stack_frames(x, emb, t, blend=a.reshape(bt))    # (b, t) order = frame-major rows
            xt = self._time_stack_frames(x, emb, t)
            return torch.lerp(xt, x, a.to(x.dtype))
        xs = x.reshape(bt // t, t, c, h, w).transpose(1, 2)          # b c t h w (view)
        xt = self.time_stack(xs, emb.reshape(bt // t, t, *emb.shape[1:]))
        out = self.time_mixer(x_spatial=xs, x_temporal=xt, image_only_indicator=image_only_indicator)
        return out.transpose(1, 2).reshape(bt, c, h, w)
