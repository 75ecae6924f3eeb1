"""First-stage autoencoder of the SVD pipeline: KL image encoder + temporal video decoder.

Reference: sgm/modules/diffusionmodules/model.py:52-201 (Normalize, Upsample, Downsample, ResnetBlock,
AttnBlock), :487-601 (Encoder), :604-748 (Decoder); sgm/modules/autoencoding/temporal_ae.py:16-81
(VideoResBlock), :84-108 (AE3DConv), :111-180 (VideoBlock), :291-347 (VideoDecoder);
sgm/modules/autoencoding/regularizers/__init__.py:13-31 + sgm/modules/distributions/distributions.py:24-60
(DiagonalGaussianRegularizer); sgm/models/autoencoder.py:102-219 (AutoencodingEngine.encode/decode);
sgm/models/diffusion.py:193-226 (decode_first_stage / encode_first_stage, scale_factor 0.18215,
configs/test/svd_f_est_ctrl_simp1.yaml:5-6, :124-159). State-dict keys are the reference's.

What differs from the reference graph (results identical, tested against its golden outputs):
* GroupNorm+SiLU is one fused op; a convolution's bias is folded into the next norm (`chan_bias`) or added
  together with the residual in one pass;
* the temporal ResBlock (Conv3d kernel (3,1,1) on `b c t h w`) is evaluated on the frame-major tensor the
  spatial layers produce: temporal GroupNorm statistics are taken with a frame stride and each (3,1,1)
  convolution is one 1x1 convolution over the channel-stacked (t-1 | t | t+1) input, so the two
  `(b t) c h w <-> b c t h w` permutes per block never happen;
* the mid-block attention is single-head with D = C (512): scores of one frame chunk are materialised
  (288 GB of HBM make S x S fp32 affordable: 340 MB per frame at 72x128 latents) by a library GEMM and
  normalised in place by the fused scale+softmax kernel.
"""
import math
import os
from typing import Iterable, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .layers import ResBlock, conv_no_bias, temporal_conv3_stacked, timestep_embedding
from .transformer import VideoTransformerBlock


class Normalize(nn.GroupNorm):
    """GroupNorm(32, eps 1e-6, affine) (model.py:52-55); fused SiLU / channel bias on request."""

    def __init__(self, in_channels, num_groups=32):
        super().__init__(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)

    def forward(self, x, silu: bool = False, chan_bias=None):
        return ops.group_norm(x, self.num_groups, self.weight, self.bias, self.eps, silu=silu, chan_bias=chan_bias)


def nonlinearity(x):
    return x * torch.sigmoid(x)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        return self.conv(x) if self.with_conv else x


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def forward(self, x):
        if self.with_conv:
            return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))
        return F.avg_pool2d(x, kernel_size=2, stride=2)


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels, self.use_conv_shortcut = out_channels, conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if temb_channels > 0:
            self.temb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
            else:
                self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x, temb=None):
        h = conv_no_bias(self.conv1, self.norm1(x, silu=True))
        e = self.conv1.bias.float()[None].expand(x.shape[0], -1)          # conv1's bias rides inside norm2
        if temb is not None:
            e = e + self.temb_proj(nonlinearity(temb)).float()
        h = self.norm2(h, silu=True, chan_bias=e.contiguous())
        h = conv_no_bias(self.conv2, self.dropout(h))
        if self.in_channels != self.out_channels:
            sk = self.conv_shortcut if self.use_conv_shortcut else self.nin_shortcut
            return h + conv_no_bias(sk, x, sk.bias + self.conv2.bias)
        return ops.bias_residual_add(h, self.conv2.bias, x)


class VideoResBlock(ResnetBlock):
    """Spatial ResnetBlock, then a temporal ResBlock over the frame axis, blended by sigmoid(mix_factor)
    (temporal_ae.py:16-81)."""

    def __init__(self, out_channels, *args, dropout=0.0, video_kernel_size=3, alpha=0.0, merge_strategy="learned",
                 **kwargs):
        super().__init__(out_channels=out_channels, dropout=dropout, *args, **kwargs)
        if video_kernel_size is None:
            video_kernel_size = [3, 1, 1]
        self.time_stack = ResBlock(channels=out_channels, emb_channels=0, dropout=dropout, dims=3,
                                   use_scale_shift_norm=False, use_conv=False, up=False, down=False,
                                   kernel_size=video_kernel_size, use_checkpoint=False, skip_t_emb=True)
        self.merge_strategy = merge_strategy
        if merge_strategy == "fixed":
            self.register_buffer("mix_factor", torch.Tensor([alpha]))
        elif merge_strategy == "learned":
            self.register_parameter("mix_factor", nn.Parameter(torch.Tensor([alpha])))
        else:
            raise ValueError(f"unknown merge strategy {self.merge_strategy}")

    def get_alpha(self, bs=None):
        if self.merge_strategy == "fixed":
            return self.mix_factor
        return torch.sigmoid(self.mix_factor)

    def _frames_path_ok(self):
        conv = self.time_stack.in_layers[2]
        return (isinstance(conv, nn.Conv3d) and tuple(conv.kernel_size) == (3, 1, 1) and tuple(conv.padding) == (1, 0, 0)
                and tuple(conv.stride) == (1, 1, 1))

    def _time_stack_frames(self, x, T):
        """openaimodel.py:328-354 with dims=3, skip_t_emb on x [(b T), c, h, w] (no permutes)."""
        ts = self.time_stack
        g0, g1 = ts.in_layers[0], ts.out_layers[0]
        c1, c2 = ts.in_layers[2], ts.out_layers[3]
        h3 = ops.group_norm_frames(x, T, g0.num_groups, g0.weight, g0.bias, g0.eps, silu=True, stack3=True)
        h = temporal_conv3_stacked(h3, c1, with_bias=False)
        e = c1.bias.float()[None].expand(x.shape[0], -1).contiguous()
        h3 = ops.group_norm_frames(h, T, g1.num_groups, g1.weight, g1.bias, g1.eps, silu=True, chan_bias=e, stack3=True)
        h = temporal_conv3_stacked(ts.out_layers[2](h3), c2, with_bias=False)
        return ops.bias_residual_add(h, c2.bias, x)

    def forward(self, x, temb=None, skip_video=False, timesteps=None):
        if timesteps is None:
            timesteps = self.timesteps
        x = super().forward(x, temb)
        if skip_video:
            return x
        T = int(timesteps)
        if self._frames_path_ok():
            xt = self._time_stack_frames(x, T)
        else:
            bt, c, h, w = x.shape
            x5 = x.reshape(bt // T, T, c, h, w).transpose(1, 2)
            xt = self.time_stack(x5, temb).transpose(1, 2).reshape(bt, c, h, w)
        alpha = self.get_alpha().to(x.dtype)
        return alpha * xt + (1.0 - alpha) * x


class AE3DConv(nn.Conv2d):
    """Conv2d followed by a Conv3d over (t, h, w) (temporal_ae.py:84-108)."""

    def __init__(self, in_channels, out_channels, video_kernel_size=3, *args, **kwargs):
        super().__init__(in_channels, out_channels, *args, **kwargs)
        if isinstance(video_kernel_size, Iterable):
            padding = [int(k // 2) for k in video_kernel_size]
        else:
            padding = int(video_kernel_size // 2)
        self.time_mix_conv = nn.Conv3d(out_channels, out_channels, kernel_size=video_kernel_size, padding=padding)

    def forward(self, input, timesteps, skip_video=False):
        x = super().forward(input)
        if skip_video:
            return x
        T = int(timesteps)
        tm = self.time_mix_conv
        if tuple(tm.kernel_size) == (3, 1, 1) and tuple(tm.padding) == (1, 0, 0):
            return temporal_conv3_stacked(ops._stack3(x, T), tm, with_bias=True)
        bt, c, h, w = x.shape
        x5 = x.reshape(bt // T, T, c, h, w).transpose(1, 2)
        return tm(x5).transpose(1, 2).reshape(bt, c, h, w)


class AttnBlock(nn.Module):
    """Single-head self-attention over the h*w positions of each frame with D = C (model.py:161-201)."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def _qkv_tokens(self, x):
        """norm + the three 1x1 convolutions as ONE token-major GEMM: [B, (h w), C] x [C, 3C]."""
        C = self.in_channels
        t = ops.group_norm_tokens(x, self.norm.num_groups, self.norm.weight, self.norm.bias, self.norm.eps)
        w = torch.cat([self.q.weight, self.k.weight, self.v.weight]).reshape(3 * C, C)
        b = torch.cat([self.q.bias, self.k.bias, self.v.bias])
        return F.linear(t, w, b).split(C, dim=-1)

    def attention_tokens(self, x):
        q, k, v = self._qkv_tokens(x)
        return ops.attention_wide(q, k, v)                        # [B, (h w), C]

    def attention(self, h_):
        b, c, h, w = h_.shape
        return self.attention_tokens(h_).transpose(1, 2).reshape(b, c, h, w)

    def forward(self, x, **kwargs):
        C = self.in_channels
        t = self.attention_tokens(x)
        t = F.linear(t, self.proj_out.weight.reshape(C, C), self.proj_out.bias)
        return ops.tokens_to_planes_add(t, x)


class VideoBlock(AttnBlock):
    """AttnBlock + a temporal transformer block on the attended tokens (temporal_ae.py:111-180); only built for
    time_mode "all" / "attn-only" (the shipped configs use "conv-only")."""

    def __init__(self, in_channels: int, alpha: float = 0, merge_strategy: str = "learned", attn_mode="softmax"):
        super().__init__(in_channels)
        self.time_mix_block = VideoTransformerBlock(dim=in_channels, n_heads=1, d_head=in_channels, checkpoint=False,
                                                    ff_in=True, attn_mode=attn_mode)
        time_embed_dim = self.in_channels * 4
        self.video_time_embed = nn.Sequential(nn.Linear(self.in_channels, time_embed_dim), nn.SiLU(),
                                              nn.Linear(time_embed_dim, self.in_channels))
        self.merge_strategy = merge_strategy
        if merge_strategy == "fixed":
            self.register_buffer("mix_factor", torch.Tensor([alpha]))
        elif merge_strategy == "learned":
            self.register_parameter("mix_factor", nn.Parameter(torch.Tensor([alpha])))
        else:
            raise ValueError(f"unknown merge strategy {self.merge_strategy}")

    def get_alpha(self):
        if self.merge_strategy == "fixed":
            return self.mix_factor
        return torch.sigmoid(self.mix_factor)

    def forward(self, x, timesteps, skip_video=False):
        if skip_video:
            return super().forward(x)
        C = self.in_channels
        T = int(timesteps)
        t = self.attention_tokens(x)                                   # b (h w) c
        frames = torch.arange(T, device=x.device).repeat(x.shape[0] // T)
        emb = self.video_time_embed(timestep_embedding(frames, C, repeat_only=False).to(t.dtype))
        x_mix = self.time_mix_block(t + emb[:, None, :], timesteps=T)
        alpha = self.get_alpha().to(t.dtype)
        t = alpha * t + (1.0 - alpha) * x_mix
        t = F.linear(t, self.proj_out.weight.reshape(C, C), self.proj_out.bias)
        return ops.tokens_to_planes_add(t, x)


class MemoryEfficientVideoBlock(VideoBlock):
    def __init__(self, in_channels: int, alpha: float = 0, merge_strategy: str = "learned"):
        super().__init__(in_channels, alpha=alpha, merge_strategy=merge_strategy, attn_mode="softmax-xformers")

    def forward(self, x, timesteps, skip_time_block=False):
        return super().forward(x, timesteps, skip_video=skip_time_block)


def make_attn(in_channels, attn_type="vanilla", attn_kwargs=None):
    """model.py:277-309. "vanilla-xformers" maps to the same kernel; "none" is Identity."""
    assert attn_type in ["vanilla", "vanilla-xformers", "none"], f"attn_type {attn_type} unknown"
    if attn_type == "none":
        return nn.Identity(in_channels)
    return AttnBlock(in_channels)


def make_time_attn(in_channels, attn_type="vanilla", attn_kwargs=None, alpha: float = 0, merge_strategy: str = "learned"):
    assert attn_type in ["vanilla", "vanilla-xformers"], f"attn_type {attn_type} not supported for spatio-temporal attention"
    cls = VideoBlock if attn_type == "vanilla" else MemoryEfficientVideoBlock
    return cls(in_channels, alpha=alpha, merge_strategy=merge_strategy)


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        assert not use_linear_attn, "linear attention is not used by any shipped config"
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = nn.Conv2d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.in_ch_mult = in_ch_mult
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for i_level in range(self.num_resolutions):
            for i_block in range(self.num_res_blocks):
                h = self.down[i_level].block[i_block](h, None)
                if len(self.down[i_level].attn) > 0:
                    h = self.down[i_level].attn[i_block](h)
            if i_level != self.num_resolutions - 1:
                h = self.down[i_level].downsample(h)
        h = self.mid.block_1(h, None)
        h = self.mid.attn_1(h)
        h = self.mid.block_2(h, None)
        return self.conv_out(self.norm_out(h, silu=True))


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        assert not use_linear_attn, "linear attention is not used by any shipped config"
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        make_attn_cls, make_resblock_cls, make_conv_cls = self._make_attn(), self._make_resblock(), self._make_conv()
        self.conv_in = nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = make_resblock_cls(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn_cls(block_in, attn_type=attn_type)
        self.mid.block_2 = make_resblock_cls(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(make_resblock_cls(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn_cls(block_in, attn_type=attn_type))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = make_conv_cls(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def _make_attn(self):
        return make_attn

    def _make_resblock(self):
        return ResnetBlock

    def _make_conv(self):
        return nn.Conv2d

    def get_last_layer(self, **kwargs):
        return self.conv_out.weight

    @staticmethod
    def _attn(m, h, kw):
        return m(h, **kw) if isinstance(m, VideoBlock) else m(h)

    def forward(self, z, **kwargs):
        self.last_z_shape = z.shape
        if kwargs.get("timesteps") is not None and not kwargs.get("skip_video"):
            # fp32 accuracy on the bf16 matrix pipe (round 6, svd/vae_split.py): the shipped VideoDecoder configuration in fp32 on the GPU
            # runs its 3x3 / (3,1,1) convolutions with split operands on token-major activations; anything else is the graph below
            from . import vae_split
            if vae_split.applies(self, z):
                return vae_split.decode(self, z, kwargs["timesteps"])
        res_kw = kwargs if isinstance(self.mid.block_1, VideoResBlock) else {}
        h = self.conv_in(z)
        h = self.mid.block_1(h, None, **res_kw)
        h = self._attn(self.mid.attn_1, h, kwargs)
        h = self.mid.block_2(h, None, **res_kw)
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                h = self.up[i_level].block[i_block](h, None, **res_kw)
                if len(self.up[i_level].attn) > 0:
                    h = self._attn(self.up[i_level].attn[i_block], h, kwargs)
            if i_level != 0:
                h = self.up[i_level].upsample(h)
        if self.give_pre_end:
            return h
        h = self.norm_out(h, silu=True)
        h = self.conv_out(h, **kwargs) if isinstance(self.conv_out, AE3DConv) else self.conv_out(h)
        return torch.tanh(h) if self.tanh_out else h


class VideoDecoder(Decoder):
    available_time_modes = ["all", "conv-only", "attn-only"]

    def __init__(self, *args, video_kernel_size=3, alpha: float = 0.0, merge_strategy: str = "learned",
                 time_mode: str = "conv-only", **kwargs):
        self.video_kernel_size, self.alpha, self.merge_strategy, self.time_mode = video_kernel_size, alpha, merge_strategy, time_mode
        assert time_mode in self.available_time_modes, f"time_mode parameter has to be in {self.available_time_modes}"
        super().__init__(*args, **kwargs)

    def get_last_layer(self, skip_time_mix=False, **kwargs):
        if self.time_mode == "attn-only":
            raise NotImplementedError("TODO")
        return self.conv_out.time_mix_conv.weight if not skip_time_mix else self.conv_out.weight

    def _make_attn(self):
        if self.time_mode not in ["conv-only", "only-last-conv"]:
            return lambda c, attn_type="vanilla": make_time_attn(c, attn_type=attn_type, alpha=self.alpha,
                                                                 merge_strategy=self.merge_strategy)
        return super()._make_attn()

    def _make_conv(self):
        if self.time_mode != "attn-only":
            return lambda *a, **k: AE3DConv(*a, video_kernel_size=self.video_kernel_size, **k)
        return nn.Conv2d

    def _make_resblock(self):
        if self.time_mode not in ["attn-only", "only-last-conv"]:
            return lambda **k: VideoResBlock(video_kernel_size=self.video_kernel_size, alpha=self.alpha,
                                             merge_strategy=self.merge_strategy, **k)
        return super()._make_resblock()


class DiagonalGaussianDistribution:
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std, self.var = torch.exp(0.5 * self.logvar), torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self):
        # the reference draws on the CPU with the global generator and moves the noise (distributions.py:37-41)
        return self.mean + self.std * torch.randn(self.mean.shape).to(device=self.parameters.device)

    def mode(self):
        return self.mean

    def kl(self):
        if self.deterministic:
            return torch.Tensor([0.0])
        return 0.5 * torch.sum(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=[1, 2, 3])


class DiagonalGaussianRegularizer(nn.Module):
    def __init__(self, sample: bool = True):
        super().__init__()
        self.sample = sample

    def get_trainable_parameters(self):
        yield from ()

    def forward(self, z):
        posterior = DiagonalGaussianDistribution(z)
        z = posterior.sample() if self.sample else posterior.mode()
        kl = posterior.kl()
        return z, {"kl_loss": torch.sum(kl) / kl.shape[0]}


def _instantiate(cfg, table):
    if isinstance(cfg, nn.Module):
        return cfg
    name = cfg["target"].rsplit(".", 1)[-1]
    params = dict(cfg.get("params", {}) or {})
    if cfg["target"].startswith("sgm.") and name in table:      # the reference's dotted names, without needing the drop-in on sys.path
        return table[name](**params)
    from .schedule import get_obj_from_str
    return get_obj_from_str(cfg["target"])(**params)


class AutoencodingEngine(nn.Module):
    """encode / decode surface of sgm/models/autoencoder.py:102-219 (the training half — losses, discriminator,
    EMA, Lightning hooks — is out of scope). Configs are the YAML dicts of
    configs/test/svd_f_est_ctrl_simp1.yaml:124-159 or ready modules."""
    _TABLE = {"Encoder": Encoder, "Decoder": Decoder, "VideoDecoder": VideoDecoder,
              "DiagonalGaussianRegularizer": DiagonalGaussianRegularizer}

    def __init__(self, *args, encoder_config, decoder_config, loss_config=None, regularizer_config=None, **kwargs):
        super().__init__()
        self.encoder = _instantiate(encoder_config, self._TABLE)
        self.decoder = _instantiate(decoder_config, self._TABLE)
        self.loss = nn.Identity() if loss_config is None else _instantiate(loss_config, self._TABLE)
        self.regularization = (DiagonalGaussianRegularizer() if regularizer_config is None
                               else _instantiate(regularizer_config, self._TABLE))

    def get_last_layer(self):
        return self.decoder.get_last_layer()

    def encode(self, x, return_reg_log: bool = False, unregularized: bool = False):
        z = self.encoder(x)
        if unregularized:
            return z, dict()
        z, reg_log = self.regularization(z)
        return (z, reg_log) if return_reg_log else z

    def decode(self, z, **kwargs):
        return self.decoder(z, **kwargs)

    def forward(self, x, **additional_decode_kwargs):
        z, reg_log = self.encode(x, return_reg_log=True)
        return z, self.decode(z, **additional_decode_kwargs), reg_log


def _decoder_in(first_stage_model, dtype):
    """A copy of the decoder in `dtype`, made once per version of its parameters and kept on the model object (not a submodule: the
    state dict stays the reference's)."""
    import copy
    dec = first_stage_model.decoder
    key = tuple((p.data_ptr(), p._version) for p in dec.parameters())
    hit = first_stage_model.__dict__.get("_mvi_decoder_copies", {}).get(dtype)
    if hit is None or hit[0] != key:
        red = copy.deepcopy(dec).to(dtype).eval()
        # what an autocast run keeps in fp32 stays in fp32 here: the affine parameters of the norms (a rounded per-channel gain is a
        # systematic error that no spatial average removes: 1.7 x the reference's bf16-autocast error with them rounded) and the
        # blending factors; the norm kernels take fp32 parameters beside reduced-precision activations anyway
        src = dict(dec.named_parameters())
        for name, p in red.named_parameters():
            mod = red.get_submodule(name.rsplit(".", 1)[0]) if "." in name else red
            if isinstance(mod, nn.GroupNorm) or name.endswith("mix_factor"):
                p.data = src[name].detach().float().clone()
        hit = (key, red)
        first_stage_model.__dict__.setdefault("_mvi_decoder_copies", {})[dtype] = hit
    return hit[1]


DECODE_DTYPE = {"": None, "fp32": None, "bf16": torch.bfloat16, "f16": torch.float16}[os.environ.get("MVI_VAE_DECODE_DTYPE", "")]


@torch.no_grad()
def decode_first_stage(first_stage_model, z, scale_factor: float = 0.18215, en_and_decode_n_samples_a_time: Optional[int] = None,
                       dtype: Optional[torch.dtype] = None):
    """sgm/models/diffusion.py:193-212: unscale, decode in chunks of n frames (a chunk is one "video" for the
    temporal layers), fp32 (disable_first_stage_autocast: the reference's recipe and the default here).
    dtype = torch.bfloat16 / torch.float16 (or MVI_VAE_DECODE_DTYPE=bf16): an opt-in of this package — the decoder's weights and
    activations in that type, GroupNorm statistics and the softmax in fp32 as everywhere, fp32 frames returned. At 14 x 576x1024 the
    fp32 LIBRARY decode is bound by the fp32 matrix rate (0.92 - 1.26 s, a quarter of a 25-step sample); in bf16 its error against the
    reference's fp32 frames is that of the reference's OWN bf16-autocast decode (tests/test_vae_gpu.py, budget in tests/golden/vae_full.npz).
    Since round 6 the DEFAULT itself (dtype None: fp32 contract, 1e-4) runs its convolutions on the bf16 matrix pipe with split operands
    (svd/vae_split.py, entered from VideoDecoder.forward; MVI_VAE_SPLIT=0 for the library path)."""
    dtype = DECODE_DTYPE if dtype is None else dtype
    split_mode = None
    if dtype in (None, torch.float32) or not z.is_cuda:
        dec_mod, cast = None, None
    else:
        # round 6: where the split-operand walk applies (the shipped VideoDecoder), the reduced-precision decode is that walk with one
        # rounded value per convolution operand — residual stream and norms stay fp32, no copy of the decoder, no library convolution
        from . import vae_split
        if vae_split.applies(first_stage_model.decoder, z.float()) and isinstance(first_stage_model.decoder, VideoDecoder):
            split_mode, dec_mod, cast = {torch.bfloat16: "bf16", torch.float16: "f16"}[dtype], None, None
        else:
            dec_mod, cast = _decoder_in(first_stage_model, dtype), dtype
    z = 1.0 / scale_factor * z
    n = z.shape[0] if en_and_decode_n_samples_a_time is None else en_and_decode_n_samples_a_time
    outs = []
    for r in range(math.ceil(z.shape[0] / n)):
        zc = z[r * n:(r + 1) * n]
        kw = {"timesteps": len(zc)} if isinstance(first_stage_model.decoder, VideoDecoder) else {}
        if split_mode is not None:
            from . import vae_split
            outs.append(vae_split.decode(first_stage_model.decoder, zc.float(), kw["timesteps"], mode=split_mode))
        elif dec_mod is None:
            outs.append(first_stage_model.decode(zc, **kw))
        else:
            outs.append(dec_mod(zc.to(cast), **kw).float())
    return torch.cat(outs, dim=0)


@torch.no_grad()
def encode_first_stage(first_stage_model, x, scale_factor: float = 0.18215, en_and_decode_n_samples_a_time: Optional[int] = None):
    """sgm/models/diffusion.py:214-226."""
    n = x.shape[0] if en_and_decode_n_samples_a_time is None else en_and_decode_n_samples_a_time
    outs = [first_stage_model.encode(x[r * n:(r + 1) * n]) for r in range(math.ceil(x.shape[0] / n))]
    return scale_factor * torch.cat(outs, dim=0)
