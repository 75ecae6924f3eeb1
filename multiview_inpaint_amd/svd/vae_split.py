"""The first-stage VideoDecoder at fp32 ACCURACY off the fp32 matrix path (round 6).

The reference decodes with autocast disabled (`disable_first_stage_autocast: True`, configs/test/svd_f_est_ctrl_simp1.yaml:6;
sgm/models/diffusion.py:194-212): ~94 TFLOP of fp32 convolutions per 14-frame 576x1024 decode, 0.92 s on the fp32 matrix instructions
(a quarter of a 25-step sample). Here every 3x3 and (3,1,1) convolution of the decoder (sgm/modules/diffusionmodules/model.py:604-748,
sgm/modules/autoencoding/temporal_ae.py:16-81, :291-347) runs on the bf16 matrix pipe with SPLIT OPERANDS —
    x . w ~= x_hi . w_hi + x_hi . w_lo + x_lo . w_hi,   v = v_hi + v_lo in bf16 (16 mantissa bits), fp32 accumulation
(csrc/linear_n320.hip, mvi_conv3x3_split3_f32: one implicit GEMM with a three times longer contraction) — and the activations stay
TOKEN-MAJOR fp32 [N, H W, C] between them: GroupNorm statistics in fp32 as everywhere, the apply pass writes the (hi | lo) halves the
next convolution reads (mvi_groupnorm_silu_tok2tok_split), residual adds and blends in fp32. Error against the reference's fp32 frames:
~2e-5 of the output scale (the dropped x_lo . w_lo terms), inside the 1e-4 bar of tests/test_vae_gpu.py — the SAME contract as the fp32
decode, unlike the opt-in bf16 / f16 decode of svd/vae.py.

What stays with the libraries in fp32 (small): conv_in (4 -> 512 at 72 x 128), the 1x1 shortcuts, the mid-block attention's
projections, conv_out (128 -> 3) and its (3,1,1) time mix. Module and parameter names are the reference's; this file only walks them.
"""
import os

import torch
import torch.nn.functional as F

from . import ops

SPLIT_DECODE = os.environ.get("MVI_VAE_SPLIT", "1") != "0"      # 0: the fp32 library path of svd/vae.py (same-box A/B)
_w3_cache = {}


def _w3(weight, mode):
    """split3_weight(weight, mode), once per parameter version and operand mode."""
    import weakref
    from . import hip_ops
    key = (id(weight), mode)
    ver = (weight.data_ptr(), weight._version, weight.dtype, weight.device, int(hip_ops._lib.lib().mvi_conv3x3_n320_k_order(-1)))
    hit = _w3_cache.get(key)
    if hit is None or hit[0]() is not weight or hit[1] != ver:
        hit = (weakref.ref(weight, lambda _r, k=key: _w3_cache.pop(k, None)), ver, hip_ops.split3_weight(weight, mode))
        _w3_cache[key] = hit
    return hit[2]


def applies(decoder, z):
    """The split path serves the shipped configuration: VideoDecoder, time_mode "conv-only" (VideoResBlocks with (3,1,1) time stacks,
    plain AttnBlock in the middle, AE3DConv at the end), fp32 parameters, a GPU latent, no autograd."""
    from . import vae as V
    if not (SPLIT_DECODE and isinstance(decoder, V.VideoDecoder) and z.is_cuda and z.dtype == torch.float32 and not torch.is_grad_enabled()
            and decoder.conv_in.weight.dtype == torch.float32 and type(decoder.mid.attn_1) is V.AttnBlock
            and isinstance(decoder.conv_out, V.AE3DConv) and not decoder.give_pre_end):
        return False
    blocks = [decoder.mid.block_1, decoder.mid.block_2] + [b for up in decoder.up for b in up.block]
    return (all(isinstance(b, V.VideoResBlock) and b._frames_path_ok() and b.conv1.in_channels % 64 == 0 and b.conv1.out_channels % 64 == 0
                and not (b.in_channels != b.out_channels and b.use_conv_shortcut) for b in blocks)
            and all(len(up.attn) == 0 for up in decoder.up))


def _fire_hooks(module, h, N, H, W):
    """Forward hooks registered on a submodule this path steps over (the parity tests probe mid.attn_1 and up[1].block[2]) still see
    its output, as the [N, C, H, W] view of the token-major tensor."""
    if module._forward_hooks:
        out = h.view(N, H, W, -1).permute(0, 3, 1, 2)
        for hook in list(module._forward_hooks.values()):
            hook(module, (), out)


def _gn(h, norm, mode, silu=True, chan_bias=None, frames=1):
    from . import hip_ops
    return hip_ops.group_norm_split(h, norm.num_groups, norm.weight, norm.bias, norm.eps, silu, chan_bias=chan_bias, frames=frames, mode=mode)


def _resblock(blk, h, N, H, W, T, alpha, mode):
    """vae.VideoResBlock.forward on token-major fp32 h [N, H W, C_in] -> [N, H W, C_out]."""
    from . import hip_ops
    S, Co = H * W, blk.out_channels
    # spatial ResnetBlock (model.py:96-158): norm1-SiLU-conv1, norm2(+conv1 bias)-SiLU-conv2, + shortcut
    c = hip_ops.conv_split3(_gn(h, blk.norm1, mode), _w3(blk.conv1.weight, mode), N, H, W, Co, mode=mode).view(N, S, Co)
    e = blk.conv1.bias.float()[None].expand(N, -1).contiguous()
    c = hip_ops.conv_split3(_gn(c, blk.norm2, mode, chan_bias=e), _w3(blk.conv2.weight, mode), N, H, W, Co, mode=mode).view(N, S, Co)
    if blk.in_channels != blk.out_channels:
        sk = blk.nin_shortcut
        x = F.linear(h, sk.weight.reshape(Co, blk.in_channels), sk.bias)
        x = hip_ops.rows_axpb(x, c, blk.conv2.bias)                   # shortcut + (conv2 + bias), one pass, into c's storage
    else:
        x = hip_ops.rows_axpb(h, c, blk.conv2.bias)                   # h + (conv2 + bias)
    # temporal ResBlock over the frame axis (temporal_ae.py:41-54 -> openaimodel.py:328-354 with dims = 3, skip_t_emb), blended
    ts = blk.time_stack
    g0, g1, c1, c2 = ts.in_layers[0], ts.out_layers[0], ts.in_layers[2], ts.out_layers[3]
    ct = hip_ops.conv_split3(_gn(x, g0, mode, frames=T), _w3(c1.weight, mode), N // T, T, S, Co, taps=3, mode=mode).view(N, S, Co)
    e = c1.bias.float()[None].expand(N, -1).contiguous()
    ct = hip_ops.conv_split3(_gn(ct, g1, mode, chan_bias=e, frames=T), _w3(c2.weight, mode), N // T, T, S, Co, taps=3, mode=mode).view(N, S, Co)
    # alpha * (x + ct + b) + (1 - alpha) * x = x + alpha * (ct + b)
    return hip_ops.rows_axpb(x, ct, c2.bias, alpha=alpha)


def _attn(blk, h):
    """vae.AttnBlock.forward (model.py:161-201: single head, D = C) on token-major fp32 h [N, S, C]."""
    C = blk.in_channels
    t = ops.group_norm_tok2tok(h, blk.norm.num_groups, blk.norm.weight, blk.norm.bias, blk.norm.eps)
    w = torch.cat([blk.q.weight, blk.k.weight, blk.v.weight]).reshape(3 * C, C)
    b = torch.cat([blk.q.bias, blk.k.bias, blk.v.bias])
    q, k, v = F.linear(t, w, b).split(C, dim=-1)
    a = ops.attention_wide(q, k, v)
    return h + F.linear(a, blk.proj_out.weight.reshape(C, C), blk.proj_out.bias)


def _upsample(up, h, N, H, W, mode):
    """vae.Upsample.forward (model.py:57-71: nearest x2, then the 3x3 convolution) on tokens: the split halves are written at the LOW
    resolution and repeated (one bf16 copy), the convolution runs at the high one."""
    from . import hip_ops
    C = h.shape[-1]
    t2 = hip_ops.group_norm_split(h, 0, None, None, 0.0, False, mode=mode)            # plain split / rounding, [N, H W, 2 C or C]
    Cx = t2.shape[-1]
    t2 = t2.view(N, H, 1, W, 1, Cx).expand(N, H, 2, W, 2, Cx).reshape(N, 4 * H * W, Cx)
    c = hip_ops.conv_split3(t2, _w3(up.conv.weight, mode), N, 2 * H, 2 * W, C, mode=mode).view(N, 4 * H * W, C)
    return c.add_(up.conv.bias)                                      # (one in-place pass; the next block reads it three times anyway)


@torch.no_grad()
def decode(decoder, z, timesteps, mode="split3"):
    """VideoDecoder.forward(z, timesteps=T) (temporal_ae.py:291-347 over model.py:604-748) -> [N, 3, 8 h, 8 w] fp32.
    mode "split3" (the default decode: fp32 accuracy, split operands); "bf16" / "f16": the opt-in reduced-precision decode — the same
    walk with ONE rounded value per convolution operand (the arithmetic of an autocast convolution: rounded products, fp32 accumulation),
    residual stream, norms and attention in fp32 as here."""
    from . import layers
    N, _, H, W = z.shape
    T = int(timesteps)
    blocks = [decoder.mid.block_1, decoder.mid.block_2] + [b for up in decoder.up for b in up.block]
    alphas = torch.stack([b.get_alpha().reshape(()).float() for b in blocks]).tolist()       # ONE read-back for all blend factors
    alpha = {id(b): a for b, a in zip(blocks, alphas)}
    h = decoder.conv_in(z)
    h = h.permute(0, 2, 3, 1).reshape(N, H * W, h.shape[1]).contiguous()             # token-major from here on
    h = _resblock(decoder.mid.block_1, h, N, H, W, T, alpha[id(decoder.mid.block_1)], mode)
    _fire_hooks(decoder.mid.block_1, h, N, H, W)
    h = _attn(decoder.mid.attn_1, h)
    _fire_hooks(decoder.mid.attn_1, h, N, H, W)
    h = _resblock(decoder.mid.block_2, h, N, H, W, T, alpha[id(decoder.mid.block_2)], mode)
    _fire_hooks(decoder.mid.block_2, h, N, H, W)
    for i_level in reversed(range(decoder.num_resolutions)):
        up = decoder.up[i_level]
        for blk in up.block:
            h = _resblock(blk, h, N, H, W, T, alpha[id(blk)], mode)
            _fire_hooks(blk, h, N, H, W)
        if i_level != 0:
            h = _upsample(up.upsample, h, N, H, W, mode) if up.upsample.with_conv else \
                h.view(N, H, 1, W, 1, -1).expand(N, H, 2, W, 2, h.shape[-1]).reshape(N, 4 * H * W, -1)
            H, W = 2 * H, 2 * W
    no = decoder.norm_out
    h = ops.group_norm_tok2tok(h, no.num_groups, no.weight, no.bias, no.eps, silu=True)
    x = h.view(N, H, W, -1).permute(0, 3, 1, 2)                                          # [N, C, H, W] with channels-last strides: no copy
    co = decoder.conv_out
    y = F.conv2d(x, co.weight, co.bias, co.stride, co.padding).contiguous()
    tm = co.time_mix_conv
    if tuple(tm.kernel_size) == (3, 1, 1) and tuple(tm.padding) == (1, 0, 0):
        y = layers.temporal_conv3_stacked(ops._stack3(y, T), tm, with_bias=True)
    else:
        y = tm(y.reshape(N // T, T, *y.shape[1:]).transpose(1, 2)).transpose(1, 2).reshape(N, -1, H, W)
    return torch.tanh(y) if decoder.tanh_out else y
