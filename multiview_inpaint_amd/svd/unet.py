"""VideoUNet, its ControlNet-conditioned variant and the ControlNet itself.

Reference: sgm/modules/diffusionmodules/video_model.py:84-493 (VideoUNet),
models/csvd.py:33-115 (ControlledVideoUNet), :119-498 (ControlNet; hint stem :234-250, zero
convs :431-432, forward :434-498), checkpoint helpers :500-564. Module/parameter names match the
reference so `load_state_dict` of svd.safetensors and of ControlNet checkpoints works
(sgm/models/diffusion.py:105, models/csvd.py:522-550).

The encoder (time/label embedding, input blocks, middle block) is built once by `_Encoder` and
shared by the UNet and the ControlNet, which the reference writes out twice.
"""
import contextlib
import os
from typing import List, Optional

import torch
import torch.nn as nn

from .layers import (Downsample, TimestepEmbedSequential, Timestep, Tok, Upsample, VideoResBlock, conv_nd, linear, prepare_emb_projections,
                     norm_act, normalization, timestep_embedding, to_planes, to_tok, token_stream_ok, zero_module)
from .transformer import SpatialVideoTransformer, prepare_single_token_rows
from . import ops

STEM_CONV = os.environ.get("MVI_SVD_STEM_CONV", "1") != "0"     # csrc/stem_conv.hip for the 16-channel layers of the hint stem


def _input_conv(blk, x):
    """input_blocks[0] = TimestepEmbedSequential(conv 3x3, in_channels -> model_channels) (openaimodel.py / video_model.py): with
    8 input channels and 320 outputs in reduced precision on the GPU it is the small-channel MFMA kernel of csrc/stem_conv.hip with
    the bias fused (the library needs 0.18 ms + a 0.11 ms bias pass for a 165 MB output); anything else is the module as it is."""
    if STEM_CONV and len(blk) == 1 and type(blk[0]) is nn.Conv2d and x.is_cuda and not torch.is_grad_enabled():
        from . import hip_ops
        conv = blk[0]
        if hip_ops.stem_conv3x3_supported(conv, x):
            return hip_ops.stem_conv3x3_silu(x, conv.weight, conv.bias, silu=False, stride=conv.stride[0])
    return blk(x, None)


class _Encoder(nn.Module):
    """Everything VideoUNet and ControlNet have in common up to and including the middle block."""

    def _build_encoder(self, *, in_channels, model_channels, num_res_blocks, attention_resolutions, dropout,
                       channel_mult, conv_resample, dims, num_classes, use_checkpoint, num_heads, num_head_channels,
                       num_heads_upsample, use_scale_shift_norm, resblock_updown, transformer_depth,
                       transformer_depth_middle, context_dim, time_downup, time_context_dim, extra_ff_mix_layer,
                       use_spatial_context, merge_strategy, merge_factor, spatial_transformer_attn_type,
                       video_kernel_size, use_linear_in_transformer, adm_in_channels,
                       disable_temporal_crossattention, max_ddpm_temb_period):
        assert context_dim is not None
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        assert num_heads != -1 or num_head_channels != -1
        if isinstance(transformer_depth, int):
            transformer_depth = len(channel_mult) * [transformer_depth]
        if transformer_depth_middle is None:
            transformer_depth_middle = transformer_depth[-1]
        self.dims, self.in_channels, self.model_channels = dims, in_channels, model_channels
        self.num_res_blocks, self.attention_resolutions, self.dropout = num_res_blocks, attention_resolutions, dropout
        self.channel_mult, self.conv_resample, self.num_classes = channel_mult, conv_resample, num_classes
        self.use_checkpoint, self.num_heads, self.num_head_channels = use_checkpoint, num_heads, num_head_channels
        self.num_heads_upsample, self.context_dim, self.adm_in_channels = num_heads_upsample, context_dim, adm_in_channels
        self._transformer_depth = transformer_depth

        ted = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, ted), nn.SiLU(), linear(ted, ted))
        if num_classes is not None:
            if isinstance(num_classes, int):
                self.label_emb = nn.Embedding(num_classes, ted)
            elif num_classes == "continuous":
                self.label_emb = nn.Linear(1, ted)
            elif num_classes == "timestep":
                self.label_emb = nn.Sequential(Timestep(model_channels),
                                               nn.Sequential(linear(model_channels, ted), nn.SiLU(), linear(ted, ted)))
            elif num_classes == "sequential":
                assert adm_in_channels is not None
                self.label_emb = nn.Sequential(nn.Sequential(linear(adm_in_channels, ted), nn.SiLU(), linear(ted, ted)))
            else:
                raise ValueError()

        def heads_for(ch):
            if num_head_channels == -1:
                return num_heads, ch // num_heads
            return ch // num_head_channels, num_head_channels

        def attn(ch, depth):
            nh, dh = heads_for(ch)
            return SpatialVideoTransformer(
                ch, nh, dh, depth=depth, context_dim=context_dim, time_context_dim=time_context_dim, dropout=dropout,
                ff_in=extra_ff_mix_layer, use_spatial_context=use_spatial_context, merge_strategy=merge_strategy,
                merge_factor=merge_factor, checkpoint=use_checkpoint, use_linear=use_linear_in_transformer,
                attn_mode=spatial_transformer_attn_type, disable_self_attn=False,
                disable_temporal_crossattention=disable_temporal_crossattention,
                max_time_embed_period=max_ddpm_temb_period)

        def res(ch, out_ch, down=False, up=False):
            return VideoResBlock(merge_factor=merge_factor, merge_strategy=merge_strategy,
                                 video_kernel_size=video_kernel_size, channels=ch, emb_channels=ted, dropout=dropout,
                                 out_channels=out_ch, dims=dims, use_checkpoint=use_checkpoint,
                                 use_scale_shift_norm=use_scale_shift_norm, down=down, up=up)
        self._mk_attn, self._mk_res = attn, res

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        self._feature_size = model_channels
        chans, ch, ds = [model_channels], model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(attn(ch, transformer_depth[level]))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                self._feature_size += ch
                chans.append(ch)
            if level != len(channel_mult) - 1:
                ds *= 2
                self.input_blocks.append(TimestepEmbedSequential(
                    res(ch, ch, down=True) if resblock_updown
                    else Downsample(ch, conv_resample, dims=dims, out_channels=ch, third_down=time_downup)))
                chans.append(ch)
                self._feature_size += ch
        self.middle_block = TimestepEmbedSequential(res(ch, None), attn(ch, transformer_depth_middle), res(ch, None))
        self._feature_size += ch
        return chans, ch, ds

    def _embed(self, x, timesteps, y):
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional -> no, relax this TODO"
        wdt = self.time_embed[0].weight.dtype
        emb = self.time_embed(timestep_embedding(timesteps, self.model_channels, repeat_only=False).to(wdt))
        if self.num_classes is not None:
            assert y.shape[0] == x.shape[0]
            emb = emb + self.label_emb(y)
        return emb


_CTOR_DEFAULTS = dict(dropout=0.0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                      use_checkpoint=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                      use_scale_shift_norm=False, resblock_updown=False, transformer_depth=1,
                      transformer_depth_middle=None, context_dim=None, time_downup=False, time_context_dim=None,
                      extra_ff_mix_layer=False, use_spatial_context=False, merge_strategy="fixed", merge_factor=0.5,
                      spatial_transformer_attn_type="softmax", video_kernel_size=3, use_linear_in_transformer=False,
                      adm_in_channels=None, disable_temporal_crossattention=False, max_ddpm_temb_period=10000)


def _ctor_args(kw):
    unknown = set(kw) - set(_CTOR_DEFAULTS)
    if unknown:
        raise TypeError(f"unexpected keyword arguments {sorted(unknown)}")
    return {**_CTOR_DEFAULTS, **kw}


class VideoUNet(_Encoder):
    def __init__(self, in_channels: int, model_channels: int, out_channels: int, num_res_blocks: int,
                 attention_resolutions, **kw):
        super().__init__()
        a = _ctor_args(kw)
        self.out_channels = out_channels
        chans, ch, ds = self._build_encoder(in_channels=in_channels, model_channels=model_channels,
                                            num_res_blocks=num_res_blocks, attention_resolutions=attention_resolutions, **a)
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(a["channel_mult"]))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [self._mk_res(ch + chans.pop(), model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(self._mk_attn(ch, self._transformer_depth[level]))
                if level and i == num_res_blocks:
                    ds //= 2
                    layers.append(self._mk_res(ch, ch, up=True) if a["resblock_updown"]
                                  else Upsample(ch, a["conv_resample"], dims=a["dims"], out_channels=ch, third_up=a["time_downup"]))
                self.output_blocks.append(TimestepEmbedSequential(*layers))
                self._feature_size += ch
        self.out = nn.Sequential(normalization(ch), nn.SiLU(),
                                 zero_module(conv_nd(a["dims"], model_channels, out_channels, 3, padding=1)))
        del self._mk_attn, self._mk_res

    def forward(self, x, timesteps, context=None, y=None, time_context=None, num_video_frames=None,
                image_only_indicator=None, control: Optional[List[torch.Tensor]] = None):
        emb = self._embed(x, timesteps, y)
        prepare_emb_projections(self, emb)                # all ResBlock embedding projections of this step as one GEMM per width
        prepare_single_token_rows(self, context, num_video_frames)   # every cross-attention row to the one CLIP token, batched
        kw = dict(context=context, image_only_indicator=image_only_indicator, time_context=time_context,
                  num_video_frames=num_video_frames)
        tokens = token_stream_ok(x)                        # the residual stream between the blocks held token-major (layers.Tok)
        hs, h = [], x
        for blk in self.input_blocks:
            if blk is self.input_blocks[0]:
                h = _input_conv(blk, h)
                h = to_tok(h) if tokens else h
            else:
                h = blk(h, emb, **kw)
            hs.append(h)
        h = self.middle_block(h, emb, **kw)
        if callable(control):
            control = control()                            # residuals produced on another stream: joined here (engine.py)
        if tokens:
            return self._decode_tokens(h, hs, control, emb, kw)
        if control is not None:
            h = h + to_planes(control.pop())               # consumes the caller's list (csvd.py:79-91)
        for blk in self.output_blocks:
            skip = hs.pop()
            c = to_planes(control.pop()) if control is not None else None
            if c is not None and c.shape != skip.shape:    # (a pooled residual [N, C, 1, 1] — global_average_pooling, csvd.py:1263 — broadcasts)
                skip, c = skip + c, None
            h = blk(ops.concat_add(h, skip, c), emb, **kw)
        h = h.type(x.dtype)
        out = self.out[2](norm_act(self.out, h))
        return out

    def _decode_tokens(self, h, hs, control, emb, kw):
        """The second half of forward() on the token-major stream: `h + control.pop()`, the concatenations with the skip tensors and
        their residuals (csvd.py:79-91) as row passes — the concatenation also takes the statistics of the GroupNorm that opens the
        block it feeds — and the last norm + convolution, whose 4-channel result is returned b c h w."""
        from . import hip_ops
        from .layers import GN_STATS_FROM_TAILS

        def residual(like):
            c = control.pop()
            if not isinstance(c, Tok) and tuple(c.shape[2:]) != (like.H, like.W):
                return c, False                            # (a pooled residual [N, C, 1, 1]: broadcast on planes below)
            return to_tok(c), True

        if control is not None:
            c, same = residual(h)
            h = Tok(hip_ops.rows_fused(h.t, c.t)[0], h.H, h.W) if same else to_tok(h.planes() + c)
        for blk in self.output_blocks:
            skip = hs.pop()
            c, same = residual(skip) if control is not None else (None, True)
            if not same:
                skip, c = to_tok(skip.planes() + c), None
            first = blk[0]
            g = first.in_layers[0].num_groups if GN_STATS_FROM_TAILS and hasattr(first, "in_layers") else 0
            t, st = hip_ops.rows_fused(h.t, skip.t, base=None if c is None else c.t, concat=True, groups=g)
            h = blk(Tok(t, h.H, h.W, st), emb, **kw)
        n, conv = self.out[0], self.out[2]
        N, C, H, W = h.shape
        t = ops.group_norm_tok2tok(h.t, n.num_groups, n.weight, n.bias, n.eps, silu=True, partials=h.gn_stats(n.num_groups))
        # 320 -> 4 channels: the library's b c h w convolution behind one layout pass (254 us at 576x1024) — its channels-last form on a
        # view of the tokens is slower (318 us) and not in the shipped find-db (a process's first calls ran MIOpen's naive kernel, 3.3 ms)
        return conv(Tok(t, H, W).planes())


class ControlledVideoUNet(VideoUNet):
    """VideoUNet whose skip connections and middle output receive ControlNet residuals
    (models/csvd.py:33-115). Same parameters as VideoUNet; `control` is popped from the end."""

    def forward(self, x, timesteps, context=None, y=None, time_context=None, control=None, num_video_frames=None,
                image_only_indicator=None):
        return super().forward(x, timesteps, context=context, y=y, time_context=time_context,
                               num_video_frames=num_video_frames, image_only_indicator=image_only_indicator,
                               control=control)


class ControlNet(_Encoder):
    """Encoder copy + hint stem + zero convolutions; returns 13 residuals (12 input blocks + middle)
    for the SVD configuration (models/csvd.py:119-498)."""

    def __init__(self, in_channels: int, model_channels: int, hint_channels: int, num_res_blocks: int,
                 attention_resolutions, **kw):
        super().__init__()
        a = _ctor_args(kw)
        self.hint_channels = hint_channels
        dims = a["dims"]
        chans, ch, _ = self._build_encoder(in_channels=in_channels, model_channels=model_channels,
                                           num_res_blocks=num_res_blocks, attention_resolutions=attention_resolutions, **a)
        self.zero_convs = nn.ModuleList([self.make_zero_conv(c) for c in chans])
        widths = [(hint_channels, 16, 1), (16, 16, 1), (16, 32, 2), (32, 32, 1), (32, 96, 2), (96, 96, 1), (96, 256, 2)]
        stem = []
        for cin, cout, stride in widths:
            stem += [conv_nd(dims, cin, cout, 3, padding=1, stride=stride), nn.SiLU()]
        stem.append(zero_module(conv_nd(dims, 256, model_channels, 3, padding=1)))
        self.input_hint_block = TimestepEmbedSequential(*stem)
        self.middle_block_out = self.make_zero_conv(ch)
        del self._mk_attn, self._mk_res

    def _hint_stem(self, hint, emb, context, tokens=False):
        """input_hint_block (csvd.py:234-250): convolution, SiLU, ..., convolution. On the GPU every convolution runs
        without its bias and `silu(h + bias)` is one fused pass (the tensors are up to 528 MB at 576x1024)."""
        layers_ = list(self.input_hint_block)
        plain = all(isinstance(m, (nn.Conv2d, nn.SiLU)) for m in layers_)
        if not (plain and hint.is_cuda and not torch.is_grad_enabled()):
            g = self.input_hint_block(hint, emb, context)
            return to_tok(g) if tokens else g
        from . import hip_ops, ops
        from .layers import conv_no_bias
        h, i = hint, 0
        while i < len(layers_):
            conv = layers_[i]
            if i + 1 < len(layers_) and isinstance(layers_[i + 1], nn.SiLU):
                if STEM_CONV and hip_ops.stem_conv3x3_supported(conv, h):
                    h = hip_ops.stem_conv3x3_silu(h, conv.weight, conv.bias, stride=conv.stride[0])   # the 16- / 32-channel layers of the stem
                else:
                    h = ops.bias_silu(conv_no_bias(conv, h), conv.bias)
                i += 2
            else:
                from .layers import conv3x3_planes_via_tokens
                y = conv3x3_planes_via_tokens(conv, h, tokens_out=tokens)   # the last layer, 256 -> model_channels: the implicit-GEMM kernel
                h = conv(h) if y is None else y
                i += 1
        return to_tok(h) if tokens else h

    @contextlib.contextmanager
    def hint_cache(self):
        """Within this context the output of input_hint_block is computed once per hint instead of once per denoise step.
        The stem is convolutions and SiLU only (csvd.py:234-250) — it sees neither the timestep embedding nor the
        latent — so over the 25 steps of one sample its [b T, 320, h, w] output is the same tensor 25 times; at
        576x1024 it is the most expensive full-resolution part of the step. The cache is keyed on the hint's storage
        and in-place version and on the stem parameters' versions, never used while autograd records, and dropped at
        exit. SVDInpaintEngine.sample() enters it; a bare forward() (and bench.py's per-step metric) does not."""
        self.__dict__["_hint_slot"] = [None]
        try:
            yield self
        finally:
            self.__dict__.pop("_hint_slot", None)

    def _hint_stem_cached(self, hint, emb, context, tokens=False):
        slot = self.__dict__.get("_hint_slot")
        plain = all(isinstance(m, (nn.Conv2d, nn.SiLU)) for m in self.input_hint_block)
        if slot is None or not plain or torch.is_grad_enabled() or not torch.is_tensor(hint):
            return self._hint_stem(hint, emb, context, tokens)
        key = (hint.data_ptr(), hint._version, tuple(hint.shape), hint.dtype, bool(tokens)) + tuple(
            (p.data_ptr(), p._version) for p in self.input_hint_block.parameters())
        if slot[0] is not None and slot[0][0] == key:
            return slot[0][1]
        guided = self._hint_stem(hint, emb, context, tokens)
        slot[0] = (key, guided, hint)               # holds the hint: its address cannot be reused meanwhile
        return guided

    def make_zero_conv(self, channels):
        return TimestepEmbedSequential(zero_module(conv_nd(self.dims, channels, channels, 1, padding=0)))

    offers_token_residuals = True

    def forward(self, x, hint, timesteps, context=None, y=None, time_context=None, num_video_frames=None,
                image_only_indicator=None, tokens_out=False):
        """tokens_out (not in the reference; SVDInpaintEngine.apply_model sets it): where this pass carried its residual stream
        token-major, return the 13 residuals as they are (layers.Tok) for a ControlledVideoUNet that consumes them so — otherwise they
        are returned b c h w, as the reference's."""
        emb = self._embed(x, timesteps, y)
        prepare_emb_projections(self, emb)
        prepare_single_token_rows(self, context, num_video_frames)
        kw = dict(context=context, image_only_indicator=image_only_indicator, time_context=time_context,
                  num_video_frames=num_video_frames)
        tokens = token_stream_ok(x) and torch.is_tensor(hint)
        guided = self._hint_stem_cached(hint, emb, context, tokens)
        outs, h = [], x
        for blk, zc in zip(self.input_blocks, self.zero_convs):
            h = _input_conv(blk, h) if blk is self.input_blocks[0] else blk(h, emb, **kw)
            if guided is not None:
                if tokens:
                    from . import hip_ops                  # b c h w + tokens -> tokens: the layout change rides on the add
                    h = Tok(hip_ops.planes_add_to_tokens(h, guided.t), guided.H, guided.W)
                else:
                    h = h + guided                         # added once, after the first input block (:471-473)
                guided = None
            outs.append(zc(h, emb, context))
        h = self.middle_block(h, emb, **kw)
        outs.append(self.middle_block_out(h, emb, context))
        return outs if tokens_out else [to_planes(o) for o in outs]

    # ---- checkpoint helpers (models/csvd.py:500-564)
    @staticmethod
    def _read_state(path):
        if path.endswith("ckpt"):
            return torch.load(path, map_location="cpu")["state_dict"]
        if path.endswith("safetensors"):
            from safetensors.torch import load_file
            return load_file(path)
        raise NotImplementedError

    def _load(self, sd, path):
        missing, unexpected = self.load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")
        if missing:
            print(f"Missing Keys: {missing}")
        if unexpected:
            print(f"Unexpected Keys: {unexpected}")

    def init_from_ckpt(self, path: str) -> None:
        self._load(self._read_state(path), path)

    def init_ctrl_from_test(self, path: str) -> None:
        prefix = "control_model."
        sd = {k[len(prefix):]: v for k, v in self._read_state(path).items() if k.startswith(prefix)}
        self._load(sd, path)

    def init_from_unet(self, unet: nn.Module) -> None:
        """Copies the encoder weights of a (Controlled)VideoUNet (models/csvd.py:1062-1065)."""
        own = self.state_dict()
        src = {k: v for k, v in unet.state_dict().items() if k in own and own[k].shape == v.shape}
        self.load_state_dict(src, strict=False)

    def set_parameters_requires_grad(self):
        self.requires_grad_(True)

    def get_trainable_parameters(self):
        return [p for p in self.parameters() if p.requires_grad]

    def get_blacklist(self):
        return []
