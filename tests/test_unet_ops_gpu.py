"""GPU parity of the UNet device ops (HIP, through the C-ABI) against plain PyTorch fp32 references
computed on the CPU, and of the whole small UNet / ControlNet / sampler running on those ops against
the golden outputs of the imported reference (tests/golden/sgm_small.npz).

Tolerances (north_star: 1e-4 relative on UNet activations):
  * fp32 I/O: GroupNorm(+SiLU) and attention  -> 1e-4 relative (validation mode, fp32 everywhere);
  * bf16 / f16 I/O (production): inputs are rounded to the I/O type first and the reference is
    evaluated in fp64 on those rounded inputs; the bound is the I/O type's own rounding
    (bf16 2^-8, f16 2^-11) on the output plus P's rounding inside the MFMA kernel."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import svd_helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _strict_hip_path(monkeypatch):
    """MVI_STRICT for every test of this module: a GPU tensor that would leave the HIP path raises (svd/ops.py)."""
    from multiview_inpaint_amd.svd import ops as dev_ops
    monkeypatch.setattr(dev_ops, "STRICT", True)
DROPIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multiview_inpaint_amd", "dropin")
if DROPIN not in sys.path:
    sys.path.insert(0, DROPIN)


# Reduced-precision bars of the reference-pinned graphs: the build's error against the reference's fp32 output as a multiple of
# the reference's OWN autocast error in the same type (max norm and rms, per recorded tensor).
#   * at the full size of configs[3] (28 x 72x128 latents, millions of elements per tensor): 1.3 x. Observed in round 5, with the
#     softmax scale folded into the q weights: 1.10 x (bf16) / 1.14 x (f16) on the networks, 1.06 x / 0.89 x after two sampling steps.
#   * on the 16x16 / 32x32 latents of the small fixtures: 1.6 x. There the max norm over a few thousand elements is a noisy
#     statistic: swapping ONE kernel family for the library's moves the f16 multiple of the production-width fixture between 1.26 and
#     1.53 with no family standing out (tools/diag_c320_precision.py -> profiles/round5_diag_c320_f16.txt: shipped 1.27, library
#     convolutions 1.53, NCHW ResBlocks 1.46, exact-scale attention 1.34; rms multiples 0.94 - 1.25) — the "1.56 x in f16" of round 4
#     was that noise, not an op. Observed this round: 0.97 - 1.47.
#     Round 6: the max-norm multiple is not even REPEATABLE — the same build on the same box gave 1.35 / 1.39 / 1.54 (32x32-latent hd64
#     fixture, f16, three runs; the attention kernel of rounds 2 - 5 in the same session: 1.38 / 1.40 / 1.54) and once 1.69 in a full-suite
#     run: the library GEMMs do not sum in a fixed order. So the small fixtures hold the rms to 1.6 x (observed 0.94 - 1.25) and the max
#     norm — a statistic of a single element — to 2.0 x; the full-size fixtures (millions of elements) keep ONE bar for both.
FULL_SIZE_BAR = 1.3
SMALL_LATENT_BAR = 1.6
SMALL_LATENT_MAX_BAR = 2.0


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from multiview_inpaint_amd.svd import hip_ops
    return hip_ops


GN_SHAPES = [((4, 64, 24, 16), 32), ((2, 320, 9, 16), 32), ((3, 96, 5, 7), 32), ((2, 64, 3, 20, 12), 32),
             ((28, 320, 72, 128), 32), ((2, 320, 14, 72, 128), 32), ((28, 1280, 9, 16), 32), ((2, 32, 1, 3), 32),
             ((2, 640, 36, 64), 32)]


@pytest.mark.parametrize("shape,groups", GN_SHAPES)
@pytest.mark.parametrize("silu", [False, True])
def test_groupnorm_fp32(ops, shape, groups, silu):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g) * 2 + 0.7
    x[0, :2] += 30.0                                     # a group with mean >> std
    w, b = torch.randn(shape[1], generator=g), torch.randn(shape[1], generator=g)
    eps = 1e-6 if not silu else 1e-5
    ref = F.group_norm(x.double(), groups, w.double(), b.double(), eps)
    if silu:
        ref = F.silu(ref)
    y = ops.group_norm_silu(x.cuda(), groups, w.cuda(), b.cuda(), eps, silu)
    assert y.dtype == torch.float32 and y.shape == x.shape
    assert rel(y, ref) < 1e-5


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
def test_groupnorm_half_io(ops, dtype, tol):
    g = torch.Generator().manual_seed(5)
    for shape in [(4, 64, 24, 16), (2, 320, 14, 18, 32), (3, 96, 5, 7)]:
        x = (torch.randn(shape, generator=g) * 1.5).to(dtype)
        w, b = torch.randn(shape[1], generator=g), torch.randn(shape[1], generator=g)
        ref = F.silu(F.group_norm(x.double(), 32, w.double(), b.double(), 1e-5))
        y = ops.group_norm_silu(x.cuda(), 32, w.cuda(), b.cuda(), 1e-5, True)
        assert y.dtype == dtype
        assert rel(y, ref) < tol


def _attn_ref(q, k, v, heads):
    B, Sq, HD = q.shape
    D = HD // heads
    qh, kh, vh = (t.double().reshape(B, -1, heads, D).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(-1, -2) * D ** -0.5, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, Sq, HD)


ATTN_SHAPES = [  # B, H, Sq, Sk, D
    (6, 5, 14, 14, 64), (3, 2, 25, 25, 64), (2, 4, 64, 64, 16), (2, 2, 144, 144, 64), (1, 3, 100, 37, 32),
    (2, 5, 576, 576, 64), (1, 2, 130, 200, 64), (4, 10, 14, 1, 64)]


@pytest.mark.parametrize("B,Hh,Sq,Sk,D", ATTN_SHAPES)
def test_attention_fp32_validation_mode(ops, B, Hh, Sq, Sk, D):
    g = torch.Generator().manual_seed(B * 1000 + Sq)
    q, k, v = (torch.randn(B, s, Hh * D, generator=g) for s in (Sq, Sk, Sk))
    q[0, 0] *= 6.0                                        # a peaked row
    out = ops.attention(q.cuda(), k.cuda(), v.cuda(), Hh)
    assert ops.attention_kernel_kind(Sq, Sk, D, torch.float32) == 0
    assert rel(out, _attn_ref(q, k, v, Hh)) < 1e-4


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2e-2), (torch.float16, 3e-3)])
@pytest.mark.parametrize("B,Hh,Sq,Sk,D", [(2, 5, 576, 576, 64), (1, 2, 130, 200, 64), (2, 2, 144, 144, 64),
                                            (1, 5, 2304, 2304, 64), (3, 2, 33, 65, 64), (6, 5, 14, 14, 64)])
def test_attention_half_io(ops, dtype, tol, B, Hh, Sq, Sk, D):
    g = torch.Generator().manual_seed(Sq + Sk)
    q, k, v = (torch.randn(B, s, Hh * D, generator=g).to(dtype) for s in (Sq, Sk, Sk))
    out = ops.attention(q.cuda(), k.cuda(), v.cuda(), Hh)
    assert out.dtype == dtype
    assert ops.attention_kernel_kind(Sq, Sk, D, dtype) == (1 if Sk > 32 else 0)
    assert rel(out, _attn_ref(q, k, v, Hh)) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2e-2), (torch.float16, 3e-3)])
@pytest.mark.parametrize("B,Hh,Sq,Sk", [(1, 2, 1300, 1300), (2, 3, 1024, 300), (1, 1, 1500, 257), (1, 2, 1025, 4100)])
def test_attention_8wave_kernel_ragged_shapes(ops, dtype, tol, B, Hh, Sq, Sk):
    """The 8-wave LDS-DMA kernel (S_q >= 1024, S_k >= 256) on shapes that are multiples of nothing: a last query block of
    20 / 220 / 1 rows, a last key tile of 20 / 44 / 1 / 4 keys (rows past the end are loaded clamped and masked), the K / V
    ring wrapping 1 ... 16 times; plain and packed-QKV entries."""
    g = torch.Generator().manual_seed(Sq * 7 + Sk)
    q, k, v = (torch.randn(B, s, Hh * 64, generator=g).to(dtype) for s in (Sq, Sk, Sk))
    out = ops.attention(q.cuda(), k.cuda(), v.cuda(), Hh)
    assert out.dtype == dtype and rel(out, _attn_ref(q, k, v, Hh)) < tol
    if Sq == Sk:
        qkv = torch.cat([q, k, v], -1).cuda()
        assert torch.equal(ops.attention_packed(qkv, Hh), out)


def test_attention_8wave_kernel_f16_folded_scale_on_peaked_rows(ops):
    """f16 inputs take the 8-wave kernel with softmax scale x log2(e) folded into Q (csrc/attn_flash8.hip, kExact = false: Q is
    rounded a second time, to f16's 11 bits). Peaked rows are where that shows: logits of +-20 nats with the top keys competing.
    The bar is the one the exact bf16 form is held to on the same construction (attn_check 'peaky': 5e-3) — not the random-data bar."""
    B, Hh, S, D = 1, 2, 1536, 64
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(B, S, Hh * D, generator=g) for _ in range(3))
    q *= 6.0
    qh, kh, vh = (t.to(torch.float16) for t in (q, k, v))
    out = ops.attention(qh.cuda(), kh.cuda(), vh.cuda(), Hh)
    assert ops.attention_kernel_kind(S, S, D, torch.float16) == 1
    assert torch.isfinite(out).all()
    assert rel(out, _attn_ref(qh, kh, vh, Hh)) < 5e-3


def test_attention_8wave_kernel_repeats_safely_when_the_fixed_exponent_overflows(ops):
    """The fast path keeps the first tile's row maximum as the exponent reference for the whole row; here the logits climb
    by ~115 (in log2 units) along the key axis, far past what fp32 holds relative to that reference: the row sums
    overflow, the block votes, and the whole block is repeated with the online-softmax path. One head climbs, the other is
    ordinary (its blocks must not be disturbed), and rows of a climbing head see their maximum at different keys."""
    B, Hh, S, D = 1, 2, 1024, 64
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B, S, Hh * D, generator=g)
    k = torch.randn(B, S, Hh * D, generator=g)
    v = torch.randn(B, S, Hh * D, generator=g)
    q[:, :, 0] = 16.0                                                 # head 0, channel 0: logit = 0.125 * 16 * k = 2 k
    k[:, :, 0] = torch.linspace(0.0, 40.0, S)[None]                   # 0 ... 80 nats = 115 in log2 units
    q[:, 512:, 0] = -16.0                                             # the second half of the rows peaks at key 0 instead
    qb, kb, vb = (t.to(torch.bfloat16) for t in (q, k, v))
    out = ops.attention(qb.cuda(), kb.cuda(), vb.cuda(), Hh)
    assert torch.isfinite(out).all()
    assert rel(out, _attn_ref(qb, kb, vb, Hh)) < 2e-2


def test_attention_mfma_layout_with_exact_integers(ops):
    """Asymmetric small-integer data: every product and sum is exact in bf16/fp32, so a swapped
    row/column map or a wrong key permutation in the P.V operand shows up as a gross error."""
    B, Hh, S, D = 1, 1, 128, 64
    q = torch.zeros(B, S, D)
    k = torch.zeros(B, S, D)
    for i in range(S):
        q[0, i, i % D] = 8.0                              # query i looks at channel i % 64
        k[0, i, (3 * i + 1) % D] = 8.0
    v = (torch.arange(S)[:, None] * 2 + torch.arange(D)[None, :] % 7).float()[None]
    qb, kb, vb = (t.to(torch.bfloat16) for t in (q, k, v))
    out = ops.attention(qb.cuda(), kb.cuda(), vb.cuda(), Hh)
    assert rel(out, _attn_ref(qb, kb, vb, Hh)) < 1e-2


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "sgm_small.npz"))


def test_small_unet_on_hip_ops_matches_reference_golden(G):
    """fp32 end to end on the GPU: HIP GroupNorm+SiLU and HIP attention inside the real module graph."""
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from models.csvd import ControlNet, ControlledVideoUNet
    unet = VideoUNet(**H.SMALL_UNET).eval()
    unet.load_state_dict(H.seeded_state_dict(unet, 11))
    cunet = ControlledVideoUNet(**H.SMALL_UNET).eval()
    cunet.load_state_dict(H.seeded_state_dict(cunet, 11))
    cnet = ControlNet(**H.SMALL_CTRL).eval()
    cnet.load_state_dict(H.seeded_state_dict(cnet, 12))
    unet, cunet, cnet = unet.cuda(), cunet.cuda(), cnet.cuda()
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in H.seeded_inputs(21).items()}
    kw = dict(num_video_frames=H.T_FRAMES, image_only_indicator=inp["image_only_indicator"])
    xin = torch.cat([inp["x"], inp["concat"]], 1)
    tt = 0.25 * inp["sigma"].log()
    with torch.no_grad():
        y = unet(xin, tt, inp["crossattn"], inp["vector"], **kw)
        ctrls = cnet(xin, inp["control_hint"], tt, inp["crossattn"], inp["vector"], **kw)
        yc = cunet(xin, tt, inp["crossattn"], inp["vector"], control=list(ctrls), **kw)
    assert y.is_contiguous() and rel(y, torch.tensor(G["unet_out"])) < 1e-4
    for i, c in enumerate(ctrls):
        assert rel(c, torch.tensor(G[f"ctrl_{i}"])) < 1e-4, i
    assert rel(yc, torch.tensor(G["cunet_out"])) < 1e-4


def test_sample_loop_on_gpu_matches_reference_golden_with_and_without_hint_cache(G):
    """The 5-step Euler / linear-guidance trajectory of the reference (tests/golden/sgm_small.npz, made by the imported
    reference) through SVDInpaintEngine on the GPU in fp32 — the HIP kernels inside the sampler loop, conditioning doubled
    once, ControlNet hint stem cached per sample — and the same loop without the cache: same result within 1e-4."""
    from models.csvd import ControlNet, ControlledVideoUNet, SVDInpaintEngine
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.util import instantiate_from_config
    T = H.T_FRAMES
    cunet = ControlledVideoUNet(**H.SMALL_UNET).eval()
    cunet.load_state_dict(H.seeded_state_dict(cunet, 11))
    cnet = ControlNet(**H.SMALL_CTRL).eval()
    cnet.load_state_dict(H.seeded_state_dict(cnet, 12))
    one = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in H.seeded_inputs(22, cfg_doubled=False).items()}
    sampler = instantiate_from_config({
        "target": "sgm.modules.diffusionmodules.sampling.EulerEDMSampler",
        "params": {"num_steps": 5, "device": "cuda",
                   "discretization_config": {"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization",
                                             "params": {"sigma_max": 700.0}},
                   "guider_config": {"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                                     "params": {"max_scale": 2.5, "min_scale": 1.0, "num_frames": T,
                                                "additional_cond_keys": ["control_hint"]}}}})
    eng = SVDInpaintEngine(cunet.cuda(), cnet.cuda(), Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}),
                           sampler, control_scales=[1.0] * 5)
    c = dict(crossattn=one["crossattn"], vector=one["vector"], concat=one["concat"], control_hint=one["control_hint"])
    uc = dict(crossattn=torch.zeros_like(one["crossattn"]), vector=torch.zeros_like(one["vector"]),
              concat=torch.zeros_like(one["concat"]), control_hint=one["control_hint"])
    kw = dict(num_video_frames=T, image_only_indicator=one["image_only_indicator"])
    fn = lambda x, sigma, cc: eng.denoise(x, sigma, cc, **kw)
    calls, stem = [0], cnet._hint_stem

    def counted(*a, **k):
        calls[0] += 1
        return stem(*a, **k)
    cnet._hint_stem = counted
    try:
        with torch.no_grad():
            plain = sampler(fn, one["x"].clone(), c, uc=uc)
            n_plain, calls[0] = calls[0], 0
            with cnet.hint_cache():
                cached = sampler(fn, one["x"].clone(), c, uc=uc)
    finally:
        del cnet._hint_stem
    assert n_plain == 5 and calls[0] == 1
    # not torch.equal: the vendor GEMM / convolution kernels are not run-to-run deterministic (split-K atomics); the CPU
    # test (tests/test_sgm_cpu.py) shows the cached trajectory is bit-identical where the arithmetic is deterministic
    d = rel(cached, plain)
    print(f"cached vs uncached sample on the GPU: rel {d:.2e}")
    assert d < 1e-4
    dp, dc = rel(plain, torch.tensor(G["sample_final"])), rel(cached, torch.tensor(G["sample_final"]))
    print(f"5-step fp32 trajectory on the HIP ops vs the reference's golden: rel {dp:.2e} (plain), {dc:.2e} (hint cache)")
    assert dp < 1e-4 and dc < 1e-4                 # north_star: 1e-4 rel on UNet activations


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2.0 ** -6)])
def test_cross_attention_rows_to_the_single_clip_token_are_batched(c320_nets, dtype, tol):
    """svd/transformer.py prepare_single_token_rows: `to_out(to_v(context))` of every cross-attention layer that meets the ONE CLIP token
    (attention.py:332-336 with S_k = 1: the value row, video_attention.py:250-254 for the temporal blocks' first-frame token) computed per
    network call as one GEMM over the concatenated to_v weights + one batched GEMM per width, against the per-layer rows: every eligible
    layer is in the table, its row equals the layer's own to a rounding, the network output is unchanged to the rounding of those rows,
    and changing a weight rebuilds the plan."""
    from multiview_inpaint_amd.svd import layers as LY
    from multiview_inpaint_amd.svd import transformer as TR
    cunet = c320_nets.get("cunet", dtype)
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in H.seeded_inputs(53, hw=H.LATENT_HW320, cfg=H.SMALL_UNET320).items()}
    kw = dict(num_video_frames=H.T_FRAMES, image_only_indicator=inp["image_only_indicator"])
    xin = torch.cat([inp["x"], inp["concat"]], 1).to(dtype)
    tt = 0.25 * inp["sigma"].log()
    ctx, vec = inp["crossattn"].to(dtype), inp["vector"].to(dtype)
    layers_s = [b.attn2 for b in cunet.modules() if isinstance(b, TR.BasicTransformerBlock)]
    layers_t = [v.attn2 for v in cunet.modules() if isinstance(v, TR.VideoTransformerBlock)]
    old = LY.CONV_N320_MIN_BLOCKS, TR.BATCHED_TOKEN_ROWS
    try:
        LY.CONV_N320_MIN_BLOCKS = 1
        outs = {}
        for mode in (False, True):
            TR.BATCHED_TOKEN_ROWS = mode
            del TR._row_tables[:]
            with torch.no_grad():
                outs[mode] = cunet(xin, tt, ctx, vec, **kw).float()
            torch.cuda.synchronize()
            assert len(TR._row_tables) == (2 if mode else 0)
        tables = {(base is not None): (c, table) for c, _, table, base, _ in TR._row_tables}
        for temporal, mods in ((False, layers_s), (True, layers_t)):
            c, table = tables[temporal]
            assert c.shape[0] == (ctx.shape[0] // H.T_FRAMES if temporal else ctx.shape[0]) and set(table) == {id(a) for a in mods} and len(mods) >= 3
            with torch.no_grad():
                for a in mods:
                    own = a.to_out(a.to_v(c)).float()
                    assert (table[id(a)].float() - own).abs().max().item() <= tol * max(1.0, own.abs().max().item())
        assert rel(outs[True], outs[False].double()) < (1e-4 if dtype == torch.float32 else 2e-2)
        # a changed parameter: the plan is rebuilt, the rows follow
        with torch.no_grad():
            layers_s[0].to_out[0].bias.add_(1.0)
            y2 = cunet(xin, tt, ctx, vec, **kw)
            c, table = next((c, t) for c, _, t, base, _ in TR._row_tables if base is None)
            own = layers_s[0].to_out(layers_s[0].to_v(c)).float()
            layers_s[0].to_out[0].bias.sub_(1.0)
        assert (table[id(layers_s[0])].float() - own).abs().max().item() <= tol * max(1.0, own.abs().max().item()) and torch.isfinite(y2).all()
    finally:
        LY.CONV_N320_MIN_BLOCKS, TR.BATCHED_TOKEN_ROWS = old
        del TR._row_tables[:]


@pytest.mark.parametrize("pooled", [False, True])
def test_engine_hands_the_control_residuals_over_as_tokens(c320_nets, pooled):
    """SVDInpaintEngine.apply_model (models/csvd.py:1240-1269) with the token-major residual stream: the ControlNet's 13 residuals reach the
    ControlledVideoUNet as layers.Tok (`tokens_out`), scaled by control_scales != 1 (csvd.py:1262) on the way and — pooled — averaged
    over the image (`global_average_pooling`, :1263-1264: a [N, C, 1, 1] residual the token UNet adds by broadcast on planes).
    One bf16 denoise step at production width on a 16x16 latent equals the same step with `b c h w` between the blocks
    (MVI_SVD_TOKEN_STREAM=0) to the rounding of the tails. One stream: the side-stream mode is only safe with the pinned GEMM set
    (svd/engine.py) — bench.py runs the same hand-over there."""
    from models.csvd import ControlNet, ControlledVideoUNet, SVDInpaintEngine
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from multiview_inpaint_amd.svd import engine as EG
    from multiview_inpaint_amd.svd import hip_ops
    from multiview_inpaint_amd.svd import layers as LY
    dtype = torch.bfloat16
    cunet, cnet = c320_nets.get("cunet", dtype), c320_nets.get("cnet", dtype)
    n_res = len(cnet.zero_convs) + 1
    eng = SVDInpaintEngine(cunet, cnet, Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}),
                           None, control_scales=[0.5 + 0.1 * i for i in range(n_res)], global_average_pooling=pooled)
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in H.seeded_inputs(53, hw=H.LATENT_HW320, cfg=H.SMALL_UNET320).items()}
    cond = dict(crossattn=inp["crossattn"].to(dtype), vector=inp["vector"].to(dtype), concat=inp["concat"].to(dtype), control_hint=inp["control_hint"].to(dtype))
    kw = dict(num_video_frames=H.T_FRAMES, image_only_indicator=inp["image_only_indicator"])
    old = LY.CONV_N320_MIN_BLOCKS, LY.TOKEN_STREAM, EG.TWO_STREAMS
    outs = {}
    try:
        LY.CONV_N320_MIN_BLOCKS = 1
        for stream in (True, False):
            for two in (False,):
                LY.TOKEN_STREAM, EG.TWO_STREAMS = stream, two
                hip_ops.PROFILE = [] if not two else None         # (per-op events and the side stream exclude each other)
                with torch.no_grad():
                    outs[stream, two] = eng.denoise(inp["x"], inp["sigma"], cond, **kw).float()
                torch.cuda.synchronize()
                if not two:
                    kinds = [rec[0] for rec in hip_ops.PROFILE]
                    # tokens: the 13 residuals cross without a layout pass (the first convolutions' and, pooled, the residuals' own remain)
                    assert (kinds.count("rows_concat") > 0) == stream and (kinds.count("concat_add") > 0) == (not stream), sorted(set(kinds))
                    if stream and not pooled:
                        assert kinds.count("planes_to_tokens") == 2 and kinds.count("tokens_to_planes_add") == 1, kinds
    finally:
        LY.CONV_N320_MIN_BLOCKS, LY.TOKEN_STREAM, EG.TWO_STREAMS = old
        hip_ops.PROFILE = None
    ref = outs[False, False]
    assert torch.isfinite(ref).all() and ref.abs().max() > 0
    for key, y in outs.items():
        assert rel(y, ref.double()) < 2e-2, (key, rel(y, ref.double()))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-6), (torch.bfloat16, 2e-2)])
def test_batched_embedding_projections_equal_the_per_block_ones(dtype, tol):
    """svd/layers.py prepare_emb_projections: one GEMM per output width for every ResBlock's Linear(SiLU(emb)) (+ the bias of
    the convolution in front) against the per-block path, on the small UNet and ControlNet: the table is filled and used, the
    outputs agree (fp32: to the order of summation of two GEMM kernels; bf16: a few roundings of the projection flip), and
    changing a projection's weights rebuilds the plan."""
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from multiview_inpaint_amd.svd import layers
    unet = VideoUNet(**H.SMALL_UNET).eval()
    unet.load_state_dict(H.seeded_state_dict(unet, 11))
    unet = unet.cuda().to(dtype)
    inp = {k: (v.cuda().to(dtype) if torch.is_tensor(v) and v.is_floating_point() else (v.cuda() if torch.is_tensor(v) else v))
           for k, v in H.seeded_inputs(21).items()}
    xin = torch.cat([inp["x"], inp["concat"]], 1)
    tt = 0.25 * inp["sigma"].float().log()
    kw = dict(num_video_frames=H.T_FRAMES, image_only_indicator=inp["image_only_indicator"])

    def run():
        with torch.no_grad():
            return unet(xin, tt, inp["crossattn"], inp["vector"], **kw).float()
    layers._emb_tables.clear()
    layers.BATCHED_EMB = True
    try:
        on = run()
        assert layers._emb_tables and len(layers._emb_tables[0][2]) >= 8       # every (Video)ResBlock's projection is in the table
        n_plans = len(layers._emb_plans)
        layers.BATCHED_EMB = False
        layers._emb_tables.clear()
        off = run()
        assert not layers._emb_tables
        assert rel(on, off.double()) < tol
        layers.BATCHED_EMB = True
        # (a block with more than one channel per GroupNorm group: with one, the norm removes a per-channel offset entirely)
        blk = max((m for m in unet.modules() if isinstance(m, layers.ResBlock)), key=lambda m: m.out_channels)
        with torch.no_grad():                                                   # a new parameter version: the plan must follow
            blk.emb_layers[1].bias.add_(torch.randn(blk.emb_layers[1].bias.shape, generator=torch.Generator().manual_seed(5)).cuda().to(dtype) * 2.0)
        on2 = run()
        layers.BATCHED_EMB = False
        off2 = run()
        assert rel(on2, off2.double()) < tol and rel(on2, on.double()) > 20 * max(rel(on2, off2.double()), 1e-7) and len(layers._emb_plans) == n_plans
    finally:
        layers.BATCHED_EMB = True


@pytest.mark.parametrize("dtype,bound", [(torch.bfloat16, 8e-2), (torch.float16, 1e-2)])
def test_small_unet_reduced_precision_autocast_error_is_reported(G, dtype, bound):
    """Reduced precision: bf16 autocast (what bench.py runs) and fp16 autocast (the reference's own recipe, csvd.py:27-31:
    the half-precision HIP ops — MFMA attention with the scale folded into Q, the K = 320 kernels, the norms — on f16 data).
    Not a 1e-4 claim — the measured error against the fp32 reference golden is bounded loosely and printed."""
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    unet = VideoUNet(**H.SMALL_UNET).eval()
    unet.load_state_dict(H.seeded_state_dict(unet, 11))
    unet = unet.cuda()
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in H.seeded_inputs(21).items()}
    xin = torch.cat([inp["x"], inp["concat"]], 1)
    with torch.no_grad(), torch.autocast("cuda", dtype=dtype):
        y = unet(xin, 0.25 * inp["sigma"].log(), inp["crossattn"], inp["vector"], num_video_frames=H.T_FRAMES,
                 image_only_indicator=inp["image_only_indicator"])
    assert torch.isfinite(y).all()
    e = rel(y.float(), torch.tensor(G["unet_out"]))
    print(f"{dtype}-autocast small-UNet relative error vs fp32 reference: {e:.3e}")
    assert e < bound


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 1.0 / 128)])
@pytest.mark.parametrize("bo,T,S,Hh,D", [(2, 14, 37, 5, 64), (1, 25, 8, 2, 16), (3, 3, 128, 2, 32)])
def test_attention_temporal_matches_regrouped_attention(ops, dtype, tol, bo, T, S, Hh, D):
    """Strided frame attention == regroup '(b t) s c -> (b s) t c', attend, regroup back (video_attention.py:115-140)."""
    g = torch.Generator().manual_seed(T * 100 + S)
    q, k, v = (torch.randn(bo * T, S, Hh * D, generator=g).to(dtype) for _ in range(3))
    out = ops.attention_temporal(q.cuda(), k.cuda(), v.cuda(), Hh, T)
    rg = lambda t: t.reshape(bo, T, S, Hh * D).transpose(1, 2).reshape(bo * S, T, Hh * D)
    ref = _attn_ref(rg(q), rg(k), rg(v), Hh).reshape(bo, S, T, Hh * D).transpose(1, 2).reshape(bo * T, S, Hh * D)
    assert out.dtype == dtype and rel(out, ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.0 / 128)])
@pytest.mark.parametrize("b,T,C,H,W", [(2, 14, 320, 9, 16), (1, 3, 64, 5, 7), (2, 14, 320, 72, 128), (3, 5, 96, 8, 8)])
def test_groupnorm_frames_matches_5d_groupnorm(ops, dtype, tol, b, T, C, H, W):
    """Temporal GroupNorm on the (b t) c h w layout == GroupNorm of the permuted b c t h w tensor."""
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(b * T, C, H, W, generator=g) * 1.7 + 0.3).to(dtype)
    w, bb = torch.randn(C, generator=g), torch.randn(C, generator=g)
    x5 = x.double().reshape(b, T, C, H, W).transpose(1, 2)
    ref = F.silu(F.group_norm(x5, 32, w.double(), bb.double(), 1e-5)).transpose(1, 2).reshape(b * T, C, H, W)
    y = ops.group_norm_silu_frames(x.cuda(), T, 32, w.cuda(), bb.cuda(), 1e-5, True)
    assert y.dtype == dtype and y.shape == x.shape and rel(y, ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
def test_geglu(ops, dtype, tol):
    g = torch.Generator().manual_seed(3)
    for shape in [(7, 33, 2 * 1280), (2, 5, 2 * 128), (28 * 576, 2 * 2560)]:
        h = (torch.randn(shape, generator=g) * 2).to(dtype)
        a, gate = h.double().chunk(2, -1)
        ref = a * F.gelu(gate)
        out = ops.geglu(h.cuda())
        assert out.dtype == dtype and out.shape == ref.shape and rel(out, ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.0 / 128)])
@pytest.mark.parametrize("b,T,C,H,W", [(2, 14, 320, 9, 16), (1, 1, 64, 5, 8), (2, 3, 96, 8, 8), (1, 2, 64, 3, 5)])
def test_groupnorm_frames_fused_bias_and_stacked_output(ops, dtype, tol, b, T, C, H, W):
    """chan_bias is added before the statistics; stack3 lays (t-1 | t | t+1) on the channel axis with zero ends."""
    from multiview_inpaint_amd.svd import ops as dev_ops
    g = torch.Generator().manual_seed(C + T)
    x = (torch.randn(b * T, C, H, W, generator=g) * 1.3).to(dtype)
    w, bb, cb = torch.randn(C, generator=g), torch.randn(C, generator=g), torch.randn(b * T, C, generator=g)
    ref = dev_ops.group_norm_frames(x.double().cpu(), T, 32, w.double(), bb.double(), 1e-5, silu=True, chan_bias=cb.double(), stack3=True)
    y = ops.group_norm_silu_frames(x.cuda(), T, 32, w.cuda(), bb.cuda(), 1e-5, True, chan_bias=cb.cuda(), stack3=True)
    assert y.shape == (b * T, 3 * C, H, W) and rel(y, ref) < tol
    yv = y.reshape(b, T, 3, C, H, W)
    assert (yv[:, 0, 0] == 0).all() and (yv[:, -1, 2] == 0).all()
    # plain layout with the bias only
    ref2 = dev_ops.group_norm(x.double().cpu(), 32, w.double(), bb.double(), 1e-5, silu=False, chan_bias=cb.double())
    y2 = ops.group_norm_silu(x.cuda(), 32, w.cuda(), bb.cuda(), 1e-5, False, chan_bias=cb.cuda())
    assert rel(y2, ref2) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.0 / 128)])
def test_groupnorm_one_launch_cluster_form(ops, dtype, tol):
    """Plain groups larger than one block's registers are normalised by up to 8 blocks that exchange partial moments and
    meet at per-group counters (gn_cluster_kernel): ragged pieces (7 per group in fp32, 4 in bf16) with the fused bias, an odd
    number of groups (surplus blocks of the XCD round-robin exit), the level-0 decoder shape; results against fp64, the
    counters re-armed (all zero) after every call, no block ever timed out, a second call gives the same bits."""
    from multiview_inpaint_amd.svd import ops as dev_ops
    g = torch.Generator().manual_seed(77)
    x = (torch.randn(3, 64, 90, 104, generator=g) * 1.3 + 0.2).to(dtype)
    x[0, :5] += 25.0
    w, bb, cb = torch.randn(64, generator=g), torch.randn(64, generator=g), torch.randn(3, 64, generator=g)
    ref = dev_ops.group_norm(x.double(), 4, w.double(), bb.double(), 1e-5, silu=True, chan_bias=cb.double())
    xs = x.cuda()
    y = ops.group_norm_silu(xs, 4, w.cuda(), bb.cuda(), 1e-5, True, chan_bias=cb.cuda())
    assert rel(y, ref) < tol
    assert torch.equal(y, ops.group_norm_silu(xs, 4, w.cuda(), bb.cuda(), 1e-5, True, chan_bias=cb.cuda()))
    for shape, groups in (((5, 320, 72, 128), 32), ((3, 640, 72, 128), 32), ((1, 96, 64, 96), 3)):
        x2 = (torch.randn(shape, generator=g) * 2 + 0.7).to(dtype)
        w2, b2 = torch.randn(shape[1], generator=g), torch.randn(shape[1], generator=g)
        ref2 = F.group_norm(x2.double(), groups, w2.double(), b2.double(), 1e-6)
        assert rel(ops.group_norm_silu(x2.cuda(), groups, w2.cuda(), b2.cuda(), 1e-6, False), ref2) < tol
    torch.cuda.synchronize()
    assert ops.groupnorm_cluster_timeouts() == 0
    assert all(int(buf.view(torch.int32).abs().sum()) == 0 for buf in ops._gn_sync.values())


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
def test_ff_geglu_fused_projection(ops, dtype, tol):
    """(x W_v^T + b_v) * gelu(x W_g^T + b_g) in one MFMA kernel (csrc/ff_geglu.hip, K = 320) against fp64 on the rounded
    inputs: rows that are multiples of nothing (the kernel stores whole 256-row blocks into a padded buffer), 2 ... 95 output
    steps (odd and even: the two accumulator sets swap every step), no bias, x as a strided view, gate values deep in both
    tails of the GELU; and never less accurate than library GEMM + geglu, which rounds the [rows, 2 inner] intermediate."""
    from multiview_inpaint_amd.svd import ops as dev_ops
    g = torch.Generator().manual_seed(41)
    for rows, inner, with_bias, strided in [(1000, 1280, True, False), (777, 64, False, False), (70001, 1280, True, True),
                                            (256, 3040, True, False), (33, 96, True, False)]:
        K = 320
        wide = (torch.randn(rows, 2 * K if strided else K, generator=g) * 1.2).to(dtype)
        x = wide[:, :K]
        w = (torch.randn(2 * inner, K, generator=g) * K ** -0.5).to(dtype)
        w[inner:inner + 4] *= 6.0                                    # gates of +-10 and beyond: gelu(g) -> g and -> -0
        b = (torch.randn(2 * inner, generator=g) * 0.3).to(dtype) if with_bias else None
        h = F.linear(x.double(), w.double(), None if b is None else b.double())
        ref = h[:, :inner] * F.gelu(h[:, inner:])
        xs = wide.cuda()[:, :K]                                      # a view with row stride 640 when `strided`
        assert ops.ff_geglu_supported(K, inner, dtype) and xs.is_contiguous() != strided
        y = ops.ff_geglu(xs, w.cuda(), None if b is None else b.cuda())
        assert y.shape == (rows, inner) and y.dtype == dtype and torch.isfinite(y).all()
        e_fused = rel(y, ref)
        e_lib = rel(ops.geglu(F.linear(xs, w.cuda(), None if b is None else b.cuda())), ref)
        assert e_fused < tol and e_fused <= 1.05 * e_lib + 1e-4, (rows, inner, e_fused, e_lib)
    # dispatch: enough rows -> the fused kernel (op log), few rows or another K -> library GEMM + geglu; same numbers either way
    x = (torch.randn(2, dev_ops.FF_GEGLU_MIN_ROWS // 2, 320, generator=g)).to(dtype).cuda()
    w = (torch.randn(256, 320, generator=g) * 320 ** -0.5).to(dtype).cuda()
    ops.PROFILE = []
    big = dev_ops.linear_geglu(x, w, None)
    small = dev_ops.linear_geglu(x[:, :100], w, None)
    kinds = [e[0] for e in ops.PROFILE]
    ops.PROFILE = None
    assert big.shape == (2, dev_ops.FF_GEGLU_MIN_ROWS // 2, 128) and kinds == ["ff_geglu", "geglu"], kinds
    assert rel(big[:, :100], small.double()) < 2 * tol
    assert not ops.ff_geglu_supported(640, 1280, dtype) and not ops.ff_geglu_supported(320, 1280, torch.float32)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
def test_ff_geglu_long_contraction(ops, dtype, tol):
    """mvi_ff_geglu_n320 (round 5): the GEGLU projection at K = 640 / 1280 in csrc/linear_n320.hip's frame (160 value + 160 gate
    columns per block, gated in registers) against fp64 on the rounded inputs — ragged row counts, one to sixteen column groups, no
    bias, a strided x, gates deep in both GELU tails — and never less accurate than the library GEMM + geglu it replaces (which
    rounds the [rows, 2 inner] intermediate); the dispatch of ops.linear_geglu takes it for enough rows at those K only."""
    from multiview_inpaint_amd.svd import ops as dev_ops
    g = torch.Generator().manual_seed(43)
    for rows, K, inner, with_bias, strided in [(1000, 640, 2560, True, False), (777, 640, 160, False, False), (4099, 1280, 320, True, True),
                                               (256, 128, 480, True, False), (33, 1280, 5120, False, False)]:
        wide = (torch.randn(rows, 2 * K if strided else K, generator=g) * 1.2).to(dtype)
        x = wide[:, :K]
        w = (torch.randn(2 * inner, K, generator=g) * K ** -0.5).to(dtype)
        w[inner:inner + 4] *= 6.0                                    # gates of +-10 and beyond: gelu(g) -> g and -> -0
        b = (torch.randn(2 * inner, generator=g) * 0.3).to(dtype) if with_bias else None
        h = F.linear(x.double(), w.double(), None if b is None else b.double())
        ref = h[:, :inner] * F.gelu(h[:, inner:])
        xs = wide.cuda()[:, :K]
        assert ops.ff_geglu_n320_supported(K, inner, dtype) and xs.is_contiguous() != strided
        y = ops.ff_geglu_n320(xs, w.cuda(), None if b is None else b.cuda())
        assert y.shape == (rows, inner) and y.dtype == dtype and torch.isfinite(y).all()
        e_fused = rel(y, ref)
        e_lib = rel(ops.geglu(F.linear(xs, w.cuda(), None if b is None else b.cuda())), ref)
        assert e_fused < tol and e_fused <= 1.05 * e_lib + 1e-4, (rows, K, inner, e_fused, e_lib)
    assert not ops.ff_geglu_n320_supported(640, 100, dtype) and not ops.ff_geglu_n320_supported(96, 160, dtype)
    x = torch.randn(2, dev_ops.FF_GEGLU_N320_MIN_ROWS // 2, 640, generator=g).to(dtype).cuda()
    w = (torch.randn(640, 640, generator=g) * 640 ** -0.5).to(dtype).cuda()
    ops.PROFILE = []
    big = dev_ops.linear_geglu(x, w, None)
    small = dev_ops.linear_geglu(x[:, :100], w, None)
    kinds = [e[0] for e in ops.PROFILE]
    ops.PROFILE = None
    assert big.shape == (2, dev_ops.FF_GEGLU_N320_MIN_ROWS // 2, 320) and kinds == ["ff_geglu_n320", "geglu"], kinds
    assert rel(big[:, :100], small.double()) < 2 * tol


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
def test_ff_geglu_n320_persistent_grid_equals_the_plain_grid(ops, dtype, tol, monkeypatch):
    """Round 6: launches of mvi_ff_geglu_n320 with more tiles than CUs run ONE block per CU that walks its tiles, the next tile's first W
    chunks / x rows requested under the last chunks of the tile in hand and its bias staged in LDS (csrc/linear_n320.hip kPersist;
    MVI_N320_PERSIST=0: the plain grid, read per launch). Same products in the same order: the two forms agree bit for bit — ragged row
    counts (a last tile with rows past the end), with and without bias, K = 640 / 1280 / 128-chunk-pairs — and the persistent one
    against fp64 on the rounded inputs."""
    g = torch.Generator().manual_seed(47)
    for rows, K, inner, with_bias in [(4200, 640, 2560, True), (9 * 256 + 5, 1280, 5120, False), (70000, 256, 640, True),
                                      (4700, 640, 2400, True)]:      # (285 tiles: the eight XCD shares are 35 or 36 tiles)
        x = (torch.randn(rows, K, generator=g) * 1.2).to(dtype)
        w = (torch.randn(2 * inner, K, generator=g) * K ** -0.5).to(dtype)
        b = (torch.randn(2 * inner, generator=g) * 0.3).to(dtype) if with_bias else None
        xc, wc, bc = x.cuda(), w.cuda(), None if b is None else b.cuda()
        assert -(-rows // 256) * (inner // 160) > 256                      # more tiles than the chip has CUs
        monkeypatch.setenv("MVI_N320_PERSIST", "0")
        plain = ops.ff_geglu_n320(xc, wc, bc)
        monkeypatch.setenv("MVI_N320_PERSIST", "1")
        pers = ops.ff_geglu_n320(xc, wc, bc)
        torch.cuda.synchronize()
        assert torch.equal(plain, pers), (rows, K, inner)
        if rows <= 5000:
            h = F.linear(x.double(), w.double(), None if b is None else b.double())
            assert rel(pers, h[:, :inner] * F.gelu(h[:, inner:])) < tol
    monkeypatch.delenv("MVI_N320_PERSIST")


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
def test_stem_conv3x3_silu_kernel(ops, dtype, tol):
    """silu(conv2d(x, w, b, padding=1)) for 16 output channels and <= 16 input channels on the MFMA kernel of csrc/stem_conv.hip
    (the stride-1 layers of the ControlNet hint stem at its two finest resolutions): against fp64 on the rounded inputs, 7 / 16 / 3 / 9
    input channels into 16 and 32 / 20 into 32 (the padded-channel forms of the three instantiations), images whose height and width are not multiples of the 4 x 64 tile (borders,
    ragged last tiles), with and without bias / SiLU; the dispatcher's conditions."""
    g = torch.Generator().manual_seed(53)
    for N, Cin, Cout, Hh, Ww, with_bias, silu in [(2, 7, 16, 20, 128, True, True), (1, 16, 16, 37, 72, True, True), (3, 3, 16, 4, 64, False, True),
                                                  (1, 9, 16, 9, 200, True, False), (2, 16, 16, 64, 64, True, True), (2, 32, 32, 19, 136, True, True), (2, 8, 320, 18, 72, True, False), (1, 4, 320, 9, 64, False, False),
                                                  (1, 20, 32, 8, 64, False, True)]:
        x = torch.randn(N, Cin, Hh, Ww, generator=g).to(dtype)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).to(dtype)
        b = (torch.randn(Cout, generator=g) * 0.3).to(dtype) if with_bias else None
        ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)
        ref = F.silu(ref) if silu else ref
        y = ops.stem_conv3x3_silu(x.cuda(), w.cuda(), None if b is None else b.cuda(), silu=silu)
        assert y.shape == ref.shape and y.dtype == dtype
        assert rel(y, ref) < tol
    for N, Cin, Hh, Ww in [(2, 16, 20, 128), (1, 5, 37, 80), (1, 16, 7, 16), (2, 12, 64, 256)]:          # stride 2, <= 16 -> 32: odd heights, ragged tiles
        x = torch.randn(N, Cin, Hh, Ww, generator=g).to(dtype)
        w = (torch.randn(32, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).to(dtype)
        b = (torch.randn(32, generator=g) * 0.3).to(dtype)
        ref = F.silu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1))
        y = ops.stem_conv3x3_silu(x.cuda(), w.cuda(), b.cuda(), silu=True, stride=2)
        assert y.shape == ref.shape and y.dtype == dtype
        assert rel(y, ref) < tol
    conv = torch.nn.Conv2d(7, 16, 3, padding=1).to(dtype).cuda()
    xs = torch.randn(1, 7, 8, 64, generator=g).to(dtype).cuda()
    assert ops.stem_conv3x3_supported(conv, xs)
    assert not ops.stem_conv3x3_supported(conv, xs[:, :, :, :60].contiguous())           # width not a multiple of 8
    assert ops.stem_conv3x3_supported(torch.nn.Conv2d(7, 32, 3, padding=1).to(dtype).cuda(), xs)
    assert not ops.stem_conv3x3_supported(torch.nn.Conv2d(7, 48, 3, padding=1).to(dtype).cuda(), xs)
    assert not ops.stem_conv3x3_supported(torch.nn.Conv2d(7, 16, 3, padding=1, stride=2).to(dtype).cuda(), xs)
    assert ops.stem_conv3x3_supported(torch.nn.Conv2d(7, 32, 3, padding=1, stride=2).to(dtype).cuda(), xs)
    assert not ops.stem_conv3x3_supported(conv, xs.float())


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
def test_linear_n320_with_the_add_and_layernorm_behind_it_in_the_epilogue(ops, dtype, tol):
    """mvi_linear_n320_add_layernorm — the projection that ends an attention / FeedForward layer with the residual add(s) and the NEXT
    LayerNorm in its epilogue (attention.py:544-572, video_attention.py:110-141) — against the two kernels it replaces
    (linear_n320 + add_layernorm: the residual stream s and s_pre BIT-EQUAL, the same rounding points; y within two ulps of the
    storage type, the row sums are taken in another order) and against fp64 on the same rounded inputs: with residual + broadcast
    row + s_pre (the temporal block's entry), residual only, residual + row, neither (proj_in -> norm1); ragged rows (a last block
    of 232 rows / 1 row), K = 320 and 1280, a strided x. Then through the dispatcher (ops.linear_add_layer_norm): one kernel for
    enough rows, the two-kernel form below, MVI_LN_EPILOGUE's switch."""
    from multiview_inpaint_amd.svd import ops as dev_ops
    g = torch.Generator().manual_seed(53)
    N = 320
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    for rows, K, G, with_resid, ret_pre, strided in [(1000, 1280, 4, True, True, False), (2561, 320, 0, True, False, True),
                                                     (512, 320, 2, True, False, False), (769, 1280, 0, False, False, False),
                                                     (768, 192, 3, False, True, False)]:
        wide = (torch.randn(rows, K + 64 if strided else K, generator=g) * 1.1).to(dtype)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype)
        b = (torch.randn(N, generator=g) * 0.3).to(dtype)
        lw, lb = torch.randn(N, generator=g) * 0.5 + 1.0, torch.randn(N, generator=g) * 0.2
        resid = (torch.randn(rows, N, generator=g) * 1.5).to(dtype) if with_resid else None
        row = (torch.randn(G, 1, N, generator=g)).to(dtype) if G else None
        xs = wide.cuda()[:, :K]
        args = dict(resid=None if resid is None else resid.cuda(), row=None if row is None else row.cuda(), ret_pre=ret_pre)
        y, s_, sp = ops.linear_n320_add_layer_norm(xs, w.cuda(), b.cuda(), lw.cuda(), lb.cuda(), 1e-5, **args)
        h = ops.linear_n320(xs, w.cuda(), b.cuda())
        if with_resid:
            y2, s2, sp2 = ops.add_layer_norm(args["resid"], lw.cuda(), lb.cuda(), 1e-5, h=h, row=args["row"], ret_pre=ret_pre)
        else:
            y2, s2, sp2 = ops.add_layer_norm(h, lw.cuda(), lb.cuda(), 1e-5, row=args["row"], ret_pre=ret_pre)
            s2 = h if s2 is None else s2
        assert y.shape == (rows, N) and y.dtype == dtype and s_.shape == (rows, N)
        assert torch.equal(s_, s2), (rows, K, G)
        if ret_pre:
            assert torch.equal(sp, sp2), (rows, K, G)
        else:
            assert sp is None
        scale = y2.float().abs().max().item()
        assert (y.float() - y2.float()).abs().max().item() <= 2 * ulp * scale, (rows, K, G)
        # fp64 on the same rounded inputs, the sums rounded where the storage type rounds them
        hd = F.linear(wide[:, :K].double(), w.double(), b.double()).to(dtype).double()
        sd = hd if resid is None else (resid.double() + hd).to(dtype).double()
        if row is not None:
            sd = (sd.reshape(G, rows // G, N) + row.double().reshape(G, 1, N)).reshape(rows, N).to(dtype).double()
        ref = F.layer_norm(sd, (N,), lw.double(), lb.double(), 1e-5)
        assert rel(y, ref) < tol
    # the dispatcher
    lin = torch.nn.Linear(1280, 320).to(dtype).cuda()
    norm = torch.nn.LayerNorm(320).to(dtype).cuda()
    x = torch.randn(dev_ops.FF_GEGLU_MIN_ROWS, 1280, generator=g).to(dtype).cuda()
    r = torch.randn(dev_ops.FF_GEGLU_MIN_ROWS, 320, generator=g).to(dtype).cuda()
    with torch.no_grad():
        ops.PROFILE = []
        y, s_, _ = dev_ops.linear_add_layer_norm(x, lin, r, norm)
        y3, s3, _ = dev_ops.linear_add_layer_norm(x[:1000], lin, r[:1000], norm)           # too few rows: two kernels
        kinds = [e[0] for e in ops.PROFILE]
        assert kinds == ["linear_n320_ln", "add_layernorm"], kinds
        assert rel(s_[:1000], s3.double()) < tol and rel(y[:1000], y3.double()) < tol       # (the library GEMM behind s3 sums in another order)
        dev_ops.LN_EPILOGUE = False
        try:
            ops.PROFILE = []
            y4, s4, _ = dev_ops.linear_add_layer_norm(x, lin, r, norm)
            kinds = [e[0] for e in ops.PROFILE]
        finally:
            dev_ops.LN_EPILOGUE = True
            ops.PROFILE = None
        assert kinds == ["linear_n320", "add_layernorm"], kinds
        assert torch.equal(s_, s4) and (y.float() - y4.float()).abs().max().item() <= 2 * ulp * y4.float().abs().max().item()


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
def test_linear_n320_kernel(ops, dtype, tol):
    """nn.Linear INTO 320 channels with a long contraction (csrc/linear_n320.hip: outputs stationary in accumulators, K streamed
    in chunks of 64 through an LDS ring): against fp64 and against the library GEMM, ragged rows (a last block of 232 / 1 / 44
    rows, waves that start past the end), an even and an odd number of chunks (K = 1280, 128, 192, 704), with and without bias, a
    strided x; the dispatcher takes it only for enough rows; the library keeps every other shape."""
    from multiview_inpaint_amd.svd import ops as dev_ops
    g = torch.Generator().manual_seed(47)
    for rows, K, with_bias, strided, N in [(1000, 1280, True, False, 320), (70001, 128, False, True, 320), (300, 192, True, False, 320),
                                           (2561, 704, True, True, 320), (1300, 640, True, False, 640), (513, 2560, True, True, 640),
                                           (700, 1280, False, False, 1280), (257, 192, True, False, 1920)]:      # 320 g outputs: column groups
        wide = (torch.randn(rows, K + 64 if strided else K, generator=g) * 1.2).to(dtype)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype)
        b = (torch.randn(N, generator=g) * 0.3).to(dtype) if with_bias else None
        ref = F.linear(wide[:, :K].double(), w.double(), None if b is None else b.double())
        xs = wide.cuda()[:, :K]
        assert ops.linear_n320_supported(K, N, dtype)
        y = ops.linear_n320(xs, w.cuda(), None if b is None else b.cuda())
        lib = F.linear(xs, w.cuda(), None if b is None else b.cuda())
        assert y.shape == (rows, N) and y.dtype == dtype
        assert rel(y, ref) < tol and rel(y, lib.double()) < tol
    x = torch.randn(dev_ops.FF_GEGLU_MIN_ROWS, 1280, generator=g).to(dtype).cuda()
    w = (torch.randn(320, 1280, generator=g) * 1280 ** -0.5).to(dtype).cuda()
    ops.PROFILE = []
    big = dev_ops.linear(x, w)
    small = dev_ops.linear(x[:100], w)                               # too few rows: library
    other = dev_ops.linear(x, torch.cat([w, w, w]))                  # 960 outputs: library (the column-group form is taken for 640 / 1280)
    kinds = [e[0] for e in ops.PROFILE]
    assert kinds == ["linear_n320"] and other.shape[-1] == 960
    assert rel(big[:100], small.double()) < tol
    # 640 outputs: the column-group form once its grid fills the chip (N320_GROUP_MIN_BLOCKS row blocks x groups), the library below
    w2 = torch.cat([w, w])
    ops.PROFILE = []
    wide_grid = dev_ops.linear(x[:256 * 100], w2)                    # 100 row blocks x 2 groups
    thin_grid = dev_ops.linear(x[:256 * 99], w2)                     # 198 blocks: library
    kinds = [e[0] for e in ops.PROFILE]
    ops.PROFILE = None
    assert kinds == ["linear_n320"] and rel(wide_grid[:256 * 99], thin_grid.double()) < tol
    assert rel(wide_grid[:, :320], big[:256 * 100].double()) < tol and rel(wide_grid[:, 320:], big[:256 * 100].double()) < tol
    assert ops.linear_n320_supported(1280, 640, dtype) and not ops.linear_n320_supported(1280, 600, dtype)
    assert not ops.linear_n320_supported(1000, 320, dtype) and not ops.linear_n320_supported(64, 320, dtype)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
def test_linear_k320_kernel(ops, dtype, tol):
    """nn.Linear with K = 320 on the MFMA kernel of ff_geglu (plain epilogue, 64 outputs per step): against fp64 and against the
    library GEMM (same inputs, fp32 accumulation either way: equal to the order of summation), ragged rows, 2 ... 15 steps, with
    and without bias, a strided x; the dispatcher takes it only for enough rows."""
    from multiview_inpaint_amd.svd import ops as dev_ops
    g = torch.Generator().manual_seed(43)
    for rows, N, with_bias, strided in [(1000, 960, False, False), (70001, 320, True, True), (300, 128, True, False), (513, 704, True, False)]:
        K = 320
        wide = (torch.randn(rows, 2 * K if strided else K, generator=g) * 1.2).to(dtype)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype)
        b = (torch.randn(N, generator=g) * 0.3).to(dtype) if with_bias else None
        ref = F.linear(wide[:, :K].double(), w.double(), None if b is None else b.double())
        xs = wide.cuda()[:, :K]
        assert ops.linear_k320_supported(K, N, dtype)
        y = ops.linear_k320(xs, w.cuda(), None if b is None else b.cuda())
        lib = F.linear(xs, w.cuda(), None if b is None else b.cuda())
        assert y.shape == (rows, N) and y.dtype == dtype
        assert rel(y, ref) < tol and rel(y, lib.double()) < tol
    x = torch.randn(dev_ops.FF_GEGLU_MIN_ROWS, 320, generator=g).to(dtype).cuda()
    w = (torch.randn(960, 320, generator=g) * 320 ** -0.5).to(dtype).cuda()         # (the packed q | k | v projection)
    lin = torch.nn.Linear(320, 960).to(dtype).cuda()
    ops.PROFILE = []
    big = dev_ops.linear(x, w)
    small = dev_ops.linear(x[:100], w)
    with torch.no_grad():
        viamod = dev_ops.linear_module(torch.nn.Sequential(lin, torch.nn.Dropout(0.0)).eval(), x)
    with_grad = dev_ops.linear_module(lin, x)                        # parameters that require grad under autograd: library GEMM
    kinds = [e[0] for e in ops.PROFILE]
    ops.PROFILE = None
    assert kinds == ["linear_k320", "linear_k320"] and with_grad.requires_grad
    assert rel(big[:100], small.double()) < tol and rel(viamod, with_grad.detach().double()) < tol
    # 320 -> 320 (to_out, proj_in / proj_out): both hand-written kernels apply; the dispatcher takes the output-stationary one
    ops.PROFILE = []
    sq = dev_ops.linear(x, w[:320].contiguous())
    assert [e[0] for e in ops.PROFILE] == ["linear_n320"] and rel(sq, F.linear(x, w[:320]).double()) < tol
    ops.PROFILE = None
    assert not ops.linear_k320_supported(320, 96, dtype) and not ops.linear_k320_supported(640, 640, dtype)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
def test_bias_residual_add(ops, dtype, tol):
    g = torch.Generator().manual_seed(11)
    for shape in [(4, 64, 24, 16), (3, 96, 5, 7), (2, 320, 14, 9, 16)]:
        h = torch.randn(shape, generator=g).to(dtype)
        x = torch.randn(shape, generator=g).to(dtype)
        b = torch.randn(shape[1], generator=g)
        view = (1, -1) + (1,) * (len(shape) - 2)
        for xx, bb in ((x, b), (None, b), (x, None)):
            ref = h.double() + (0 if bb is None else bb.double().view(view)) + (0 if xx is None else xx.double())
            out = ops.bias_residual_add(h.cuda(), None if bb is None else bb.cuda(), None if xx is None else xx.cuda())
            assert out.dtype == dtype and rel(out, ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
def test_bias_residual_blend(ops, dtype, tol):
    """x + (1 - alpha[n]) (h + bias) against AlphaBlender's alpha x + (1 - alpha)(x + h + bias) in fp64; rows with
    alpha == 1 (image-only frames) return x exactly."""
    g = torch.Generator().manual_seed(12)
    for shape in [(6, 64, 24, 16), (4, 96, 5, 7), (28, 320, 9, 16)]:
        h = torch.randn(shape, generator=g).to(dtype)
        x = torch.randn(shape, generator=g).to(dtype)
        b = torch.randn(shape[1], generator=g)
        a = torch.rand(shape[0], generator=g)
        a[1] = 1.0
        av = a.double().view(-1, 1, 1, 1)
        for bb in (b, None):
            xt = x.double() + h.double() + (0 if bb is None else bb.double().view(1, -1, 1, 1))
            ref = av * x.double() + (1 - av) * xt
            out = ops.bias_residual_blend(h.cuda(), None if bb is None else bb.cuda(), x.cuda(), a.cuda())
            assert out.dtype == dtype and rel(out, ref) < tol
            assert torch.equal(out[1].cpu(), x[1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_concat_add_is_bit_exact(ops, dtype):
    """cat([h, skip + ctrl], 1) in one pass: bit-identical to the two PyTorch ops (one rounding of the sum), with and
    without ctrl, vector and scalar paths (odd spatial sizes), and C1 != C2."""
    g = torch.Generator().manual_seed(13)
    for N, C1, C2, sp in [(4, 64, 64, (24, 16)), (3, 96, 32, (5, 7)), (28, 320, 320, (9, 16)), (2, 7, 5, (3, 3))]:
        h = torch.randn(N, C1, *sp, generator=g).to(dtype).cuda()
        sk = torch.randn(N, C2, *sp, generator=g).to(dtype).cuda()
        ct = torch.randn(N, C2, *sp, generator=g).to(dtype).cuda()
        assert torch.equal(ops.concat_add(h, sk, ct), torch.cat([h, sk + ct], 1))
        assert torch.equal(ops.concat_add(h, sk, None), torch.cat([h, sk], 1))
    with pytest.raises(ValueError):
        ops.concat_add(h, sk[:, :, :2], None)


LN_CASES = [(2, 36, 320), (3, 16, 640), (2, 8, 1280), (4, 6, 32), (2, 5, 64), (2, 7, 48), (1, 3, 24), (28, 2304, 640)]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
@pytest.mark.parametrize("B,S,C", LN_CASES)
@pytest.mark.parametrize("mode", ["plain", "h", "row", "h+row"])
def test_add_layernorm(ops, dtype, tol, B, S, C, mode):
    """s_pre / s are bit-exact against the op-by-op PyTorch graph (same roundings to the storage type); the
    LayerNorm output is compared with an fp64 LayerNorm of that same s."""
    if dtype == torch.float32 and C % 4 or dtype != torch.float32 and C % 8:
        pytest.skip("C not a multiple of the 16-byte vector")
    if not ops.layernorm_supported(C, dtype):
        pytest.skip("row split unsupported")
    g = torch.Generator().manual_seed(B * 1000 + S * 10 + C)
    x = (torch.randn(B, S, C, generator=g) * 1.5 + 0.3).to(dtype).cuda()
    h = (torch.randn(B, S, C, generator=g)).to(dtype).cuda() if "h" in mode else None
    row = (torch.randn(B, 1, C, generator=g)).to(dtype).cuda() if "row" in mode else None
    w, b = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    y, s, s_pre = ops.add_layer_norm(x, w, b, 1e-5, h=h, row=row, ret_pre=True)
    ref_pre = x if h is None else x + h
    ref_s = ref_pre if row is None else ref_pre + row
    assert torch.equal(s_pre, ref_pre)
    if s is None:
        assert mode == "plain"
    else:
        assert torch.equal(s, ref_s)
    ref_y = F.layer_norm(ref_s.double(), (C,), w.double(), b.double(), 1e-5)
    assert y.dtype == dtype and rel(y, ref_y) < tol


def test_add_layernorm_row_runs_and_errors(ops):
    """row broadcast over runs longer than one batch entry (the per-video cross-attention row) and the error paths."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 10, 64, generator=g).bfloat16().cuda()            # 6 = 2 videos x 3 frames
    row = torch.randn(2, 1, 64, generator=g).bfloat16().cuda()
    w, b = torch.ones(64).cuda(), torch.zeros(64).cuda()
    y, s, _ = ops.add_layer_norm(x, w, b, 1e-5, row=row)
    assert torch.equal(s, x + row.repeat_interleave(3, dim=0))
    with pytest.raises(ValueError):
        ops.add_layer_norm(x, w, b, 1e-5, row=row[:, :, :32])
    with pytest.raises(ValueError):
        ops.add_layer_norm(x, w, b, 1e-5, row=torch.zeros(7, 1, 64, device="cuda", dtype=torch.bfloat16))
    with pytest.raises(Exception):
        ops.add_layer_norm(torch.zeros(2, 3, 20, device="cuda", dtype=torch.bfloat16), torch.ones(20).cuda(), torch.zeros(20).cuda(), 1e-5)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
@pytest.mark.parametrize("with_h", [False, True])
def test_add_lerp(ops, dtype, tol, with_h):
    g = torch.Generator().manual_seed(9)
    B, S, C = 6, 37, 320
    x, h, base = ((torch.randn(B, S, C, generator=g)).to(dtype).cuda() for _ in range(3))
    alpha = torch.tensor([0.62, 1.0, 0.3, 0.0, 0.5, 0.999]).to(dtype).cuda()
    out = ops.add_lerp(x, h if with_h else None, base, alpha)
    t = x + h if with_h else x
    ref = torch.lerp(t, base, alpha.reshape(B, 1, 1))
    assert out.dtype == dtype and rel(out, ref) < tol
    assert torch.equal(out[1], base[1]) and torch.equal(out[3], t[3])          # alpha = 1 / 0 are exact


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,C,H,W", [(2, 320, 9, 16), (3, 64, 8, 8), (1, 640, 5, 8), (2, 1280, 3, 8), (2, 72, 4, 6)])
def test_tokens_to_planes_add(ops, dtype, N, C, H, W):
    g = torch.Generator().manual_seed(N + C + H)
    tok = torch.randn(N, H * W, C, generator=g).to(dtype).cuda()
    x_in = torch.randn(N, C, H, W, generator=g).to(dtype).cuda()
    out = ops.tokens_to_planes_add(tok, x_in)
    ref = tok.transpose(1, 2).reshape(N, C, H, W) + x_in
    assert out.shape == x_in.shape and torch.equal(out, ref)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
@pytest.mark.parametrize("N,C,H,W", [(2, 320, 9, 16), (3, 64, 8, 8), (2, 1280, 3, 8), (2, 96, 4, 6), (28, 320, 72, 128)])
@pytest.mark.parametrize("silu,with_bias", [(False, False), (True, True)])
def test_groupnorm_token_major_output(ops, dtype, tol, N, C, H, W, silu, with_bias):
    """GroupNorm(+chan_bias, +SiLU) written as [N, (h w), C] equals the NCHW result transposed."""
    g = torch.Generator().manual_seed(N * 7 + C + H)
    x = (torch.randn(N, C, H, W, generator=g) * 1.3 + 0.4).to(dtype)
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    cb = torch.randn(N, C, generator=g) if with_bias else None
    xf = x.double() + (cb.double()[:, :, None, None] if with_bias else 0.0)
    ref = F.group_norm(xf, 32, w.double(), b.double(), 1e-6)
    if silu:
        ref = F.silu(ref)
    y = ops.group_norm_silu_tokens(x.cuda(), 32, w.cuda(), b.cuda(), 1e-6, silu, chan_bias=None if cb is None else cb.cuda())
    assert y.shape == (N, H * W, C) and y.dtype == dtype and y.is_contiguous()
    assert rel(y, ref.flatten(2).transpose(1, 2)) < tol


# ---- packed q | k | v projections read in place (mvi_attention_*_strided) --------------------------------------------

@pytest.mark.parametrize("dtype,B,S,H,D", [(torch.float32, 3, 77, 2, 16), (torch.float32, 2, 130, 3, 64),
                                           (torch.bfloat16, 2, 300, 5, 64), (torch.float16, 3, 129, 2, 64),
                                           (torch.bfloat16, 2, 20, 4, 32)])
def test_attention_packed_equals_separate_tensors(ops, dtype, B, S, H, D):
    g = torch.Generator().manual_seed(S * H + D)
    qkv = torch.randn(B, S, 3 * H * D, generator=g).to(dtype).cuda()
    q, k, v = (t.contiguous() for t in qkv.chunk(3, dim=-1))
    want = ops.attention(q, k, v, H)
    got = ops.attention_packed(qkv, H)
    assert got.shape == want.shape and got.is_contiguous()
    assert torch.equal(got, want)                       # same kernel, same arithmetic, only the addressing differs


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
@pytest.mark.parametrize("bo,T,S,Hh", [(2, 14, 333, 5), (1, 16, 40, 10), (3, 1, 17, 5), (1, 3, 4100, 20), (2, 7, 1, 1)])
def test_attention_temporal_mfma_kernel(ops, dtype, tol, bo, T, S, Hh):
    """csrc/attn_temporal.hip (bf16 / f16, D = 64, T <= 16: every temporal attention of the SVD step) against fp64 softmax attention on
    the same rounded inputs, separate and packed q/k/v; frames 1 .. 16 (padding keys masked, padding rows never stored), more
    problems than waves (the grid-stride walk with its prefetch) and fewer, and large logits (one dominant key per row)."""
    from multiview_inpaint_amd import _lib
    D = 64
    assert _lib.lib().mvi_attention_temporal_kernel_variant(T, Hh, D, 1 if dtype == torch.bfloat16 else 2, 3 * Hh * D, Hh * D) == 1
    assert _lib.lib().mvi_attention_temporal_kernel_variant(25, Hh, D, 1, 0, 0) == 0 and _lib.lib().mvi_attention_temporal_kernel_variant(T, Hh, 32, 1, 0, 0) == 0
    g = torch.Generator().manual_seed(T * 100 + S)
    for gain in (1.0, 6.0):
        qkv = (torch.randn(bo * T, S, 3 * Hh * D, generator=g) * gain ** 0.5).to(dtype)
        q, k, v = (t.contiguous() for t in qkv.chunk(3, dim=-1))
        out_p = ops.attention_temporal_packed(qkv.cuda(), Hh, T)
        out_s = ops.attention_temporal(q.cuda(), k.cuda(), v.cuda(), Hh, T)
        torch.cuda.synchronize()
        assert torch.equal(out_p, out_s)
        rg = lambda t: t.double().reshape(bo, T, S, Hh, D).permute(0, 2, 3, 1, 4)                  # bo s h t d
        att = torch.softmax(rg(q) @ rg(k).transpose(-1, -2) * D ** -0.5, -1) @ rg(v)
        ref = att.permute(0, 3, 1, 2, 4).reshape(bo * T, S, Hh * D)
        assert out_p.dtype == dtype and (out_p.double().cpu() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_temporal_packed_equals_separate_tensors(ops, dtype):
    bo, T, S, H, D = 2, 5, 37, 3, 32
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(bo * T, S, 3 * H * D, generator=g).to(dtype).cuda()
    q, k, v = (t.contiguous() for t in qkv.chunk(3, dim=-1))
    assert torch.equal(ops.attention_temporal_packed(qkv, H, T), ops.attention_temporal(q, k, v, H, T))


def test_attention_strided_rejects_bad_strides(ops):
    from multiview_inpaint_amd import _lib
    L = _lib.lib()
    x = torch.zeros(1, 64, 3 * 64, device="cuda", dtype=torch.bfloat16)
    o = torch.zeros(1, 64, 64, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    # token stride smaller than H*D, and one that is not a 16-byte multiple
    assert L.mvi_attention_forward_strided(x.data_ptr(), x.data_ptr(), x.data_ptr(), o.data_ptr(), 1, 1, 64, 64, 64, 0.125, 1,
                                           32, 192, 64, st) != 0
    assert L.mvi_attention_forward_strided(x.data_ptr(), x.data_ptr(), x.data_ptr(), o.data_ptr(), 1, 1, 64, 64, 64, 0.125, 1,
                                           196, 196, 64, st) != 0


def test_self_attention_module_takes_the_packed_path_and_matches(ops):
    """CrossAttention self-attention at inference = one packed GEMM + in-place reads; equal to the three-projection path
    up to the GEMM library's blocking (same weights, same kernel)."""
    from multiview_inpaint_amd.svd.transformer import CrossAttention
    torch.manual_seed(3)
    m = CrossAttention(query_dim=64, heads=2, dim_head=32).cuda().eval()
    x = torch.randn(4, 50, 64, device="cuda")
    with torch.no_grad():
        got = m(x)
        q, k, v = m.to_q(x), m.to_k(x), m.to_v(x)
        want = m.to_out(ops.attention(q, k, v, 2))
        assert rel(got, want) < 1e-5
        w1, folded = m._packed_qkv_weight()
        assert not folded                                   # fp32 weights keep to_q.weight as it is
        assert m._packed_qkv_weight()[0] is w1              # cached
        m.to_k.weight.mul_(2.0)
        assert m._packed_qkv_weight()[0] is not w1          # rebuilt after an in-place weight update
        assert "_wqkv" not in m.state_dict() and len(m.state_dict()) == 5
        got_t = m.forward_temporal(x, 2)
        want_t = m.to_out(ops.attention_temporal(m.to_q(x), m.to_k(x), m.to_v(x), 2, 2))
        assert rel(got_t, want_t) < 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_softmax_scale_folded_into_the_q_weights(ops, dtype):
    """Reduced precision, sequences the 8-wave kernel serves (S >= 1024): the packed projection's q rows are
    round(dim_head^-1/2 log2(e) to_q.weight) and the kernel is told (mvi_attention_forward_strided_qlog2), so its softmax has no
    multiply per score. Against an fp64 evaluation of the module's formula from the SAME stored weights and input
    (attention.py:281-344): the folded path's error is at most 1.15 x the plain path's (to_q.weight as it is, scale applied to the
    fp32 scores) — the price of rounding the q weights a second time, printed. Shorter sequences (4-wave kernel) and the temporal
    form keep the plain weights: identical results with the switch on or off."""
    from multiview_inpaint_amd.svd import transformer as TR
    from multiview_inpaint_amd.svd.transformer import CrossAttention
    torch.manual_seed(11)
    m = CrossAttention(query_dim=320, heads=5, dim_head=64).eval()
    with torch.no_grad():
        for lin in (m.to_q, m.to_k, m.to_v):
            lin.weight.mul_(3.0)                            # logits of a few units: a softmax that is neither flat nor one-hot
    m = m.cuda().to(dtype)

    def ref(x):
        S = x.shape[1]
        xd = x.double()
        q, k, v = (xd @ w.weight.double().t() for w in (m.to_q, m.to_k, m.to_v))
        q, k, v = (t.reshape(-1, S, 5, 64).transpose(1, 2) for t in (q, k, v))
        o = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(-1, S, 320)
        return o @ m.to_out[0].weight.double().t() + m.to_out[0].bias.double()

    res = {}
    for S in (320, 1280):
        x = torch.randn(2, S, 320, device="cuda").to(dtype)
        for fold in (True, False):
            old = TR.FOLD_SCALE_INTO_WQ
            TR.FOLD_SCALE_INTO_WQ = fold
            try:
                with torch.no_grad():
                    w, q_log2 = m._packed_qkv_weight(dtype, fold=ops.attention_kernel_variant(S, S, 64, dtype) in (8, 16))
                    assert q_log2 == (fold and S == 1280)
                    if q_log2:
                        want_q = (m.to_q.weight.float() * (0.125 * 1.4426950408889634)).to(dtype)
                        assert torch.equal(w[:320], want_q) and torch.equal(w[320:640], m.to_k.weight)
                    res[S, fold] = (m(x), m.forward_temporal(x, 2))
            finally:
                TR.FOLD_SCALE_INTO_WQ = old
        assert torch.equal(res[S, True][1], res[S, False][1])                 # temporal: plain weights either way
        if S == 320:
            assert torch.equal(res[S, True][0], res[S, False][0])             # 4-wave kernel: plain weights either way
        else:
            want = ref(x)
            e_f, e_p = rel(res[S, True][0], want), rel(res[S, False][0], want)
            H.report(f"{dtype} S={S}: error of the module vs fp64: q weights carrying the scale {e_f:.2e}, plain {e_p:.2e} ({e_f / e_p:.2f} x)")
            assert e_f <= 1.15 * e_p, (e_f, e_p)


def test_temporal_conv_on_channel_stacked_input():
    """(3,1,1) temporal convolution over the channel-stacked input == the Conv3d of the reference (video_model.py:62-81)."""
    import torch.nn as nn
    from multiview_inpaint_amd.svd import layers, ops as dev_ops
    torch.manual_seed(0)
    T, C, H, W = 3, 32, 4, 6
    conv3 = nn.Conv3d(C, 48, (3, 1, 1), padding=(1, 0, 0)).cuda()
    x = torch.randn(2 * T, C, H, W, device="cuda")
    with torch.no_grad():
        want = conv3(x.reshape(2, T, C, H, W).transpose(1, 2)).transpose(1, 2).reshape(2 * T, 48, H, W)
        got = layers.temporal_conv3_stacked(dev_ops._stack3(x, T), conv3)
        assert rel(got, want) < 1e-5
        c1 = nn.Conv2d(C, 40, 1).cuda()
        assert rel(layers.conv_no_bias(c1, x, c1.bias), c1(x)) < 1e-5
        assert rel(layers.conv_no_bias(c1, x), c1(x) - c1.bias.view(1, -1, 1, 1)) < 1e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
@pytest.mark.parametrize("shape", [(2, 16, 24, 40), (3, 7, 5, 9), (1, 96, 18, 32)])
def test_bias_silu(ops, dtype, tol, shape):
    """Convolution bias + SiLU of the ControlNet hint stem in one pass."""
    g = torch.Generator().manual_seed(sum(shape))
    h = (torch.randn(*shape, generator=g) * 2).to(dtype)
    b = torch.randn(shape[1], generator=g)
    want = F.silu(h.double() + b.double().view(1, -1, 1, 1))
    got = ops.bias_silu(h.cuda().clone(), b.cuda())
    assert rel(got, want) < tol
    got = ops.bias_silu(h.cuda().clone(), None)
    assert rel(got, F.silu(h.double())) < tol


# ---- production head width (D = 64): the MFMA flash kernel against what the REFERENCE produced ------------------------

@pytest.fixture()
def strict(monkeypatch):
    """MVI_STRICT for the test body: any GPU tensor that would leave the HIP path raises (svd/ops.py)."""
    from multiview_inpaint_amd.svd import ops as dev_ops
    monkeypatch.setattr(dev_ops, "STRICT", True)
    return dev_ops


def _err(a, b):
    """(max-norm, rms) error of a against b, both relative to b's scale."""
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    d = a - b
    return float(d.abs().max() / b.abs().max()), float(d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


def test_hd64_nets_in_bf16_run_the_mfma_kernel_within_the_reference_autocast_budget(golden_dir, strict):
    """num_head_channels = 64 (configs/test/svd_f_est_ctrl_simp1.yaml:31), latent 16x16: every spatial self-attention has
    D = 64, S_k = 256 or 64 and — in bf16 — runs attn_flash_kernel INSIDE the module graph (asserted from the op log).
    Bar: the build's bf16 error against the reference's fp32 output is at most SMALL_LATENT_BAR (1.6) x the error of the reference's OWN
    reduced-precision recipe (autocast over fp32 weights, csvd.py:27-31; bf16 on the CPU) against the same fp32 output,
    per tensor, in max norm and in rms (fixture tests/golden/sgm_hd64.npz, generated from the imported reference)."""
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from models.csvd import ControlNet, ControlledVideoUNet
    from multiview_inpaint_amd.svd import hip_ops
    G64 = np.load(os.path.join(golden_dir, "sgm_hd64.npz"))
    unet = VideoUNet(**H.SMALL_UNET64).eval()
    unet.load_state_dict(H.seeded_state_dict(unet, 31))
    cunet = ControlledVideoUNet(**H.SMALL_UNET64).eval()
    cunet.load_state_dict(H.seeded_state_dict(cunet, 31))
    cnet = ControlNet(**H.SMALL_CTRL64).eval()
    cnet.load_state_dict(H.seeded_state_dict(cnet, 32))
    bf = torch.bfloat16
    unet, cunet, cnet = (m.cuda().to(bf) for m in (unet, cunet, cnet))      # the production recipe: bf16 weights + activations
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in H.seeded_inputs(41, hw=H.LATENT_HW64, cfg=H.SMALL_UNET64).items()}
    kw = dict(num_video_frames=H.T_FRAMES, image_only_indicator=inp["image_only_indicator"])
    xin = torch.cat([inp["x"], inp["concat"]], 1).to(bf)
    tt = 0.25 * inp["sigma"].log()
    ctx, vec, hint = inp["crossattn"].to(bf), inp["vector"].to(bf), inp["control_hint"].to(bf)
    hip_ops.PROFILE = []
    try:
        with torch.no_grad():
            y = unet(xin, tt, ctx, vec, **kw)
            ctrls = cnet(xin, hint, tt, ctx, vec, **kw)
            yc = cunet(xin, tt, ctx, vec, control=list(ctrls), **kw)
        torch.cuda.synchronize()
        kinds = [k for k, *_ in hip_ops.PROFILE]
    finally:
        hip_ops.PROFILE = None
    n_mfma = kinds.count("attention_mfma")
    # one spatial self-attention per SpatialVideoTransformer: 5 in the UNet (2 down, 1 middle, 2 x (1 + 1) up ...)
    assert n_mfma >= 3 * 3 and kinds.count("attention_rowtile") == 0, (n_mfma, sorted(set(kinds)))
    assert kinds.count("attention_temporal") > 0
    worst = 0.0
    for name, got in [("unet_out", y), ("cunet_out", yc)] + [(f"ctrl_{i}", c) for i, c in enumerate(ctrls)]:
        ref = G64[name + "_f32"]
        e_max, e_rms = _err(got.float(), ref)
        r_max, r_rms = _err(torch.tensor(G64[name + "_bf16ac"]), ref)
        worst = max(worst, e_max / r_max, e_rms / r_rms)
        assert e_max <= SMALL_LATENT_MAX_BAR * r_max and e_rms <= SMALL_LATENT_BAR * r_rms, (name, e_max, r_max, e_rms, r_rms)
    H.report(f"bf16 HIP path vs reference fp32: worst error ratio to the reference's own bf16-autocast error = {worst:.2f}")


@pytest.mark.parametrize("dtype,tag", [(torch.bfloat16, "bf16ac"), (torch.float16, "f16ac")])
def test_hd64_nets_on_a_32x32_latent_run_the_8_wave_kernel_within_the_reference_autocast_budget(golden_dir, strict, dtype, tag):
    """The PRODUCTION attention kernel under the reference pin: num_head_channels = 64 on a 32x32 latent, so the level-0
    spatial self-attention has S_q = S_k = 1024 and runs attn_flash8_kernel (8 waves, LDS-DMA ring — the kernel of every large
    attention of the 576 x 1024 step) INSIDE the module graph; level 1 (S = 256) runs the 4-wave kernel. Which kernel ran is
    read from mvi_attention_kernel_variant, the function the C dispatch itself uses. Reference semantics:
    sgm/modules/attention.py:281-344 (softmax(q k^T d^-1/2) v, no mask); precision recipe models/csvd.py:27-31.
    Bar, in bf16 and in f16 (the reference's own GPU recipe): the build's error against the reference's fp32 output is at most
    SMALL_LATENT_BAR (1.6) x the error of the reference's OWN autocast run in that type against the same fp32 output, per tensor, in max norm and
    in rms (fixture tests/golden/sgm_hd64.npz `L_*`, generated from the imported reference by tools/gen_golden_sgm_hd64.py)."""
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from models.csvd import ControlNet, ControlledVideoUNet
    from multiview_inpaint_amd.svd import hip_ops
    G64 = np.load(os.path.join(golden_dir, "sgm_hd64.npz"))
    unet = VideoUNet(**H.SMALL_UNET64).eval()
    unet.load_state_dict(H.seeded_state_dict(unet, 31))
    cunet = ControlledVideoUNet(**H.SMALL_UNET64).eval()
    cunet.load_state_dict(H.seeded_state_dict(cunet, 31))
    cnet = ControlNet(**H.SMALL_CTRL64).eval()
    cnet.load_state_dict(H.seeded_state_dict(cnet, 32))
    unet, cunet, cnet = (m.cuda().to(dtype) for m in (unet, cunet, cnet))
    inp = {k: (v.cuda() if torch.is_tensor(v) else v)
           for k, v in H.seeded_inputs(43, hw=H.LATENT_HW64_L, cfg=H.SMALL_UNET64).items()}
    kw = dict(num_video_frames=H.T_FRAMES, image_only_indicator=inp["image_only_indicator"])
    xin = torch.cat([inp["x"], inp["concat"]], 1).to(dtype)
    tt = 0.25 * inp["sigma"].log()
    ctx, vec, hint = inp["crossattn"].to(dtype), inp["vector"].to(dtype), inp["control_hint"].to(dtype)
    hip_ops.ATTN_VARIANTS = []
    try:
        with torch.no_grad():
            y = unet(xin, tt, ctx, vec, **kw)
            ctrls = cnet(xin, hint, tt, ctx, vec, **kw)
            yc = cunet(xin, tt, ctx, vec, control=list(ctrls), **kw)
        torch.cuda.synchronize()
        log = list(hip_ops.ATTN_VARIANTS)
    finally:
        hip_ops.ATTN_VARIANTS = None
    eight = [e for e in log if e[0] in (8, 16)]
    four = [e for e in log if e[0] == 4]
    # level 0 holds one spatial self-attention per transformer: UNet 1 down + 2 up, ControlNet 1 down, controlled UNet 1 + 2
    assert len(eight) >= 7 and all(sq == 1024 and sk == 1024 for _, sq, sk in eight), log
    assert len(four) >= 7 and all(sk == 256 for _, _, sk in four), log
    assert not [e for e in log if e[0] == 0 and e[2] > 32], log            # nothing long ran the fp32-math kernel
    worst = 0.0
    for name, got in (("L_unet_out", y), ("L_cunet_out", yc), ("L_ctrl_last", ctrls[-1])):
        ref = G64[name + "_f32"]
        e_max, e_rms = _err(got.float(), ref)
        r_max, r_rms = _err(torch.tensor(G64[name + "_" + tag]), ref)
        worst = max(worst, e_max / r_max, e_rms / r_rms)
        assert e_max <= SMALL_LATENT_MAX_BAR * r_max and e_rms <= SMALL_LATENT_BAR * r_rms, (name, e_max, r_max, e_rms, r_rms)
    H.report(f"{dtype}: 8-wave kernel x{len(eight)}, 4-wave x{len(four)}; worst error ratio to the reference's own autocast "
          f"error = {worst:.2f}")


class _C320Nets:
    """VideoUNet / ControlledVideoUNet (seed 51) and ControlNet (seed 52) of tests/golden/sgm_c320.npz: the 0.2 B seeded values per
    network are drawn once (6 s each: a third of the suite's production-width tests was drawing them again) and kept on the GPU in
    fp32; get(name, dtype) builds a fresh module over a copy in that type."""

    def __init__(self):
        self.master = {}

    def get(self, name, dt):
        from sgm.modules.diffusionmodules.video_model import VideoUNet
        from models.csvd import ControlNet, ControlledVideoUNet
        cls, cfg, seed = {"unet": (VideoUNet, H.SMALL_UNET320, 51), "cunet": (ControlledVideoUNet, H.SMALL_UNET320, 51),
                          "cnet": (ControlNet, H.SMALL_CTRL320, 52)}[name]
        key = "cnet" if name == "cnet" else "unet"             # (the two UNet classes share parameter names and seed)
        if key not in self.master:
            with torch.device("meta"):
                m = cls(**cfg)
            self.master[key] = {k: v.cuda() for k, v in H.seeded_state_dict(m, seed).items()}
        with torch.device("meta"):
            m = cls(**cfg)
        m.load_state_dict({k: (v.to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in self.master[key].items()}, strict=True, assign=True)
        return m.eval()


@pytest.fixture(scope="module")
def c320_nets():
    holder = _C320Nets()
    yield holder
    holder.master = {}
    torch.cuda.empty_cache()


@pytest.mark.parametrize("stream", ["tokens", "planes"])
@pytest.mark.parametrize("dtype,tag", [(torch.bfloat16, "bf16ac"), (torch.float16, "f16ac")])
def test_production_width_nets_run_the_round3_kernels_within_the_reference_autocast_budget(golden_dir, strict, c320_nets, dtype, tag, stream):
    """The round-3 kernels under the reference pin: model_channels = 320 and num_head_channels = 64 (the production widths,
    configs/test/svd_f_est_ctrl_simp1.yaml:18-31) on a 16x16 latent, so that INSIDE the module graphs of VideoUNet, ControlNet and
    ControlledVideoUNet the 3x3 convolutions (320 / 640 outputs, K split at this image size) and the (3,1,1) frame convolutions run
    in csrc/linear_n320.hip, VideoResBlock runs token-major (temporal GroupNorm on tokens, blend + b c h w in one pass), Upsample
    runs on tokens, and the temporal attention (T = 3, D = 64, H = 5 / 10) runs csrc/attn_temporal.hip. Which kernels ran is read
    from the op profile and from mvi_attention_temporal_kernel_variant (the function the C dispatch uses). Reference semantics:
    openaimodel.py:256-354, video_model.py:12-81, video_attention.py:110-141; precision recipe models/csvd.py:27-31.
    Bar, in bf16 and in f16 (the reference's own GPU recipe): the build's error against the reference's fp32 output is at most SMALL_LATENT_BAR (1.6) x the
    error of the reference's OWN autocast run in that type, per tensor, in max norm and in rms (fixture tests/golden/sgm_c320.npz,
    generated from the imported reference by tools/gen_golden_sgm_c320.py) — for the three final tensors AND for four
    intermediate block outputs of the UNet (first level-0 block, first level-1 block, middle block, an output block; round 4).
    stream (round 6): the residual stream between the blocks token-major (layers.Tok, the default: block tails as row passes that leave
    the next norm's statistics, no layout pass per block) or b c h w (MVI_SVD_TOKEN_STREAM=0) — the same bar for both."""
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from models.csvd import ControlNet, ControlledVideoUNet
    from multiview_inpaint_amd import _lib
    from multiview_inpaint_amd.svd import hip_ops
    from multiview_inpaint_amd.svd import layers as LY
    G = np.load(os.path.join(golden_dir, "sgm_c320.npz"))
    unet, cunet, cnet = (c320_nets.get(n, dtype) for n in ("unet", "cunet", "cnet"))
    inp = H.seeded_inputs(53, hw=H.LATENT_HW320, cfg=H.SMALL_UNET320)
    inp["image_only_indicator"][0, 1] = 1.0
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    kw = dict(num_video_frames=H.T_FRAMES, image_only_indicator=inp["image_only_indicator"])
    xin = torch.cat([inp["x"], inp["concat"]], 1).to(dtype)
    tt = 0.25 * inp["sigma"].log()
    ctx, vec, hint = inp["crossattn"].to(dtype), inp["vector"].to(dtype), inp["control_hint"].to(dtype)
    dt = 1 if dtype == torch.bfloat16 else 2
    assert _lib.lib().mvi_attention_temporal_kernel_variant(H.T_FRAMES, 5, 64, dt, 3 * 320, 320) == 1
    old = LY.CONV_N320_MIN_BLOCKS, LY.TOKEN_STREAM
    hip_ops.PROFILE = []
    try:
        LY.CONV_N320_MIN_BLOCKS = 1              # (a launch-size gate of the 576x1024 step: a 16x16 latent is far below it)
        LY.TOKEN_STREAM = stream == "tokens"
        # intermediate block outputs of the UNet (same submodule names as the reference: the state-dict keys are identical),
        # subsampled like the fixture: a wrong block that later layers wash out must not pass on the final tensors alone
        probes, handles = {}, []
        for name in H.C320_PROBES:
            handles.append(unet.get_submodule(name).register_forward_hook(
                lambda m, i, o, name=name: probes.__setitem__(name, LY.to_planes(o).detach().float()[:, ::4, ::2, ::2].contiguous())))
        with torch.no_grad():
            y = unet(xin, tt, ctx, vec, **kw)
            ctrls = cnet(xin, hint, tt, ctx, vec, **kw)
            yc = cunet(xin, tt, ctx, vec, control=list(ctrls), **kw)
        torch.cuda.synchronize()
        kinds = [rec[0] for rec in hip_ops.PROFILE]
    finally:
        LY.CONV_N320_MIN_BLOCKS, LY.TOKEN_STREAM = old
        hip_ops.PROFILE = None
        for hd in handles:
            hd.remove()
    count = lambda k: sum(1 for x in kinds if x == k)
    # per network: 5 VideoResBlocks (UNet: 2 down, middle 2, ... ) — every one token-major: 2 spatial + 2 frame convolutions each
    n_vrb = sum(1 for net in (unet, cnet, cunet) for m in net.modules() if type(m).__name__ == "VideoResBlock")
    n_svt = sum(1 for net in (unet, cnet, cunet) for m in net.modules() if type(m).__name__ == "SpatialVideoTransformer")
    n_updown = sum(1 for net in (unet, cnet, cunet) for m in net.modules() if type(m).__name__ in ("Upsample", "Downsample"))
    assert all(torch.is_tensor(c) and c.dim() == 4 for c in ctrls)                # a bare ControlNet call answers b c h w either way
    assert count("conv3t_n320") == 2 * n_vrb
    # + Upsample.conv, Downsample.op and the last convolution of the ControlNet's hint stem (256 -> 320)
    assert count("conv3x3_n320") == 2 * n_vrb + n_updown + 1 and count("attention_temporal") > 0
    if stream == "tokens":
        # every block tail a row pass: skip add + blend per VideoResBlock, `x + x_in` per transformer, the middle residual of the controlled
        # UNet; one concatenation per output block; no planes form of a tail except the ControlNet's `h + guided_hint` (planes + tokens)
        assert count("rows_blend") == n_vrb and count("rows_add") == n_vrb + n_svt + 1, kinds
        assert count("rows_concat") == len(unet.output_blocks) + len(cunet.output_blocks)
        assert count("tokens_blend_to_planes") == 0 and count("planes_add_to_tokens") == 1, kinds
        # layout passes: each network's first convolution -> tokens (the ControlNet's rides on its hint add: the hint stem's last layer
        # instead), the 13 residuals out of the bare ControlNet call and into the controlled UNet (the engine hands them over as tokens)
        # (fewer where a level has under 8 tokens: PyTorch's transpose serves those)
        assert 3 <= count("planes_to_tokens") <= 2 + 1 + len(ctrls), kinds
        assert count("groupnorm_tok2tok") == 4 * n_vrb + n_svt + 2 and count("groupnorm_tokens") == 0, kinds
    else:
        assert count("tokens_blend_to_planes") == n_vrb and count("planes_add_to_tokens") == n_vrb, kinds
        assert count("planes_to_tokens") == n_updown + 1 and count("groupnorm_tok2tok") == 3 * n_vrb and count("rows_add") == 0
    worst = 0.0
    assert set(probes) == set(H.C320_PROBES)
    for name, got in [("unet_out", y), ("cunet_out", yc), ("ctrl_last", ctrls[-1])] + [("probe_" + k, probes[k]) for k in H.C320_PROBES]:
        ref = G[name + "_f32"]
        e_max, e_rms = _err(got.float(), ref)
        r_max, r_rms = _err(torch.tensor(G[name + "_" + tag]), ref)
        worst = max(worst, e_max / r_max, e_rms / r_rms)
        assert e_max <= SMALL_LATENT_MAX_BAR * r_max and e_rms <= SMALL_LATENT_BAR * r_rms, (name, e_max, r_max, e_rms, r_rms)
    H.report(f"{dtype}: {n_vrb} token-major VideoResBlocks, {count('conv3x3_n320')} 3x3 + {count('conv3t_n320')} frame convolutions in the "
          f"implicit-GEMM kernel; worst error ratio to the reference's own autocast error = {worst:.2f}")



class _FullNets:
    """ControlledVideoUNet + ControlNet of configs[3] (1.52 B + 0.68 B parameters) with the seeded weights of
    tests/golden/sgm_full.npz: the 2.2 B seeded values are drawn ONCE per module (they took a fifth of the suite when every
    full-size test drew them again) and kept on the GPU in fp32; get(dtype) builds the modules over copies in that type."""

    def __init__(self):
        from models.csvd import ControlNet, ControlledVideoUNet
        self.specs = ((ControlledVideoUNet, H.FULL_UNET, 71), (ControlNet, H.FULL_CTRL, 72))
        self.master = None

    def _draw(self):
        if self.master is None:
            self.master = []
            for cls, cfg, seed in self.specs:
                with torch.device("meta"):
                    m = cls(**cfg)
                self.master.append({k: v.cuda() for k, v in H.seeded_state_dict(m, seed).items()})

    def get(self, dt):
        """(cunet, cnet) in dtype dt, eval mode, on the GPU: fresh modules every call (no cached derived weights carried over)."""
        self._draw()
        nets = []
        for (cls, cfg, _), sd in zip(self.specs, self.master):
            with torch.device("meta"):
                m = cls(**cfg)
            m.load_state_dict({k: (v.to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}, strict=True, assign=True)
            nets.append(m.eval())
        return nets


@pytest.fixture(scope="module")
def full_nets():
    holder = _FullNets()
    yield holder
    holder.master = None
    torch.cuda.empty_cache()

@pytest.mark.parametrize("dt,budget", [(torch.bfloat16, "budget_"), (torch.float16, "budget_f16_")])
def test_full_size_networks_match_the_reference_at_configs3_size(golden_dir, strict, full_nets, dt, budget):
    """BASELINE.json configs[3] at its REAL size against the reference itself (round 4): ControlNet + ControlledVideoUNet of
    configs/test/svd_f_est_ctrl_simp1.yaml (1.52 B + 0.68 B parameters), 14 frames on the 72 x 128 latent, CFG batch 28, bf16 on the
    HIP path (8-wave MFMA attention at S = 9216 and 2304, implicit-GEMM convolutions at every level, token-major VideoResBlocks,
    hipBLASLt / MIOpen for the rest) against tests/golden/sgm_full.npz: ONE evaluation of the imported reference on the CPU in
    fp32, plus the same evaluation under the reference's own bf16 autocast — and under f16 autocast, the reference's GPU recipe
    (yaml :214), for the f16 run of this test — as the error budget (tools/gen_golden_sgm_full.py; seeded weights and inputs,
    regenerated here bit for bit). Bar as at the production widths: error against the reference's
    fp32 output <= FULL_SIZE_BAR (1.3) x the reference's own autocast error, in max norm and in rms, for the network output, the last control
    residual and four intermediate block outputs (subsampled in the fixture)."""
    from models.csvd import ControlNet, ControlledVideoUNet
    from multiview_inpaint_amd.svd import bench_svd, hip_ops
    from multiview_inpaint_amd.svd import layers as LY
    path = os.path.join(golden_dir, "sgm_full.npz")
    assert os.path.exists(path), "tests/golden/sgm_full.npz is missing (tools/gen_golden_sgm_full.py, build container only)"
    G = np.load(path)
    assert budget + "cunet_out" in G.files, f"the fixture holds no {budget}* entries (tools/gen_golden_sgm_full.py --add-f16)"
    torch.backends.cudnn.benchmark = False
    bench_svd.use_shipped_miopen_db()
    cunet, cnet = full_nets.get(dt)
    T = H.FULL_T
    inp = H.seeded_inputs(73, T=T, hw=H.FULL_HW, cfg=H.FULL_UNET)
    inp["image_only_indicator"][0, 1] = 1.0
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    kw = dict(num_video_frames=T, image_only_indicator=inp["image_only_indicator"])
    xin = torch.cat([inp["x"], inp["concat"]], 1).to(dt)
    tt = 0.25 * inp["sigma"].log()
    ctx, vec, hint = inp["crossattn"].to(dt), inp["vector"].to(dt), inp["control_hint"].to(dt)
    probes, handles = {}, []
    for name in H.FULL_PROBES:
        handles.append(cunet.get_submodule(name).register_forward_hook(
            lambda m, i, o, name=name: probes.__setitem__(name, LY.to_planes(o).detach().float()[H.FULL_SUB].contiguous())))
    hip_ops.PROFILE = []
    try:
        with torch.no_grad():
            ctrls = cnet(xin, hint, tt, ctx, vec, **kw)
            yc = cunet(xin, tt, ctx, vec, control=list(ctrls), **kw)
        torch.cuda.synchronize()
        prof = list(hip_ops.PROFILE)
    finally:
        hip_ops.PROFILE = None
        for hd in handles:
            hd.remove()
    assert len(ctrls) == int(G["n_ctrl"]) and torch.isfinite(yc).all()
    big = [wk for kind, _, _, wk in prof if kind == "attention_mfma"]
    assert big and max(big) == 4.0 * 28 * 5 * 9216 * 9216 * 64, "the level-0 self-attention did not run the MFMA kernel"
    worst = 0.0
    for name, got in [("cunet_out", yc.float()), ("ctrl_last", ctrls[-1].float()[H.FULL_SUB])] + [("probe_" + k, probes[k]) for k in H.FULL_PROBES]:
        ref = G[name + "_f32"]
        e_max, e_rms = _err(got, ref)
        r_max, r_rms = (float(v) for v in G[budget + name])          # the reference's own autocast error in this type against the same fp32 tensor
        worst = max(worst, e_max / r_max, e_rms / r_rms)
        assert e_max <= FULL_SIZE_BAR * r_max and e_rms <= FULL_SIZE_BAR * r_rms, (name, e_max, r_max, e_rms, r_rms)
    H.report(f"full-size step in {dt} vs the reference: worst error ratio to the reference's own autocast error in that type = {worst:.2f}")


@pytest.mark.parametrize("dt,budget", [(torch.bfloat16, "budget_"), (torch.float16, "budget_f16_")])
def test_full_size_sampling_steps_match_the_reference(golden_dir, strict, full_nets, dt, budget):
    """The first two steps of the reference's sampling loop at the full size of BASELINE.json configs[3] ("SVD-xt 14-frame
    576x1024 masked-inpaint sampling"): EulerEDMSampler(num_steps = 2, sigma_max 700) + LinearPredictionGuider (1 -> 2.5,
    control_hint as an additional condition key; the guider doubles the 14 frames to the CFG batch of 28) + Denoiser(
    VScalingWithEDMcNoise) over SVDInpaintEngine.apply_model (ControlNet with the hint-stem cache -> ControlledVideoUNet) on
    the HIP path, sampler state in fp32, networks in bf16 / f16 — against the same loop of the imported reference on the CPU in
    fp32 (tools/gen_golden_sgm_full_sample.py -> tests/golden/sgm_full.npz `sample_final_f32`, subsampled [:, :, ::2, ::2]),
    with the error of the reference's own bf16 / f16 autocast loop as the budget (bar: FULL_SIZE_BAR = 1.3 x, max norm and rms)."""
    from models.csvd import ControlNet, ControlledVideoUNet, SVDInpaintEngine
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.util import instantiate_from_config
    from multiview_inpaint_amd.svd import bench_svd
    path = os.path.join(golden_dir, "sgm_full.npz")
    G = np.load(path)
    assert "sample_final_f32" in G.files and budget + "sample_final" in G.files, "run tools/gen_golden_sgm_full_sample.py (build container only)"
    torch.backends.cudnn.benchmark = False
    bench_svd.use_shipped_miopen_db()
    cunet, cnet = full_nets.get(dt)
    T = H.FULL_T
    one = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in H.seeded_inputs(74, T=T, hw=H.FULL_HW, cfg=H.FULL_UNET, cfg_doubled=False).items()}
    sampler = instantiate_from_config({
        "target": "sgm.modules.diffusionmodules.sampling.EulerEDMSampler",
        "params": {"num_steps": H.FULL_SAMPLE_STEPS, "device": "cuda",
                   "discretization_config": {"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization", "params": {"sigma_max": 700.0}},
                   "guider_config": {"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                                     "params": {"max_scale": 2.5, "min_scale": 1.0, "num_frames": T, "additional_cond_keys": ["control_hint"]}}}})
    eng = SVDInpaintEngine(cunet, cnet, Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}), sampler)
    c = dict(crossattn=one["crossattn"], vector=one["vector"], concat=one["concat"], control_hint=one["control_hint"])
    uc = dict(crossattn=torch.zeros_like(one["crossattn"]), vector=torch.zeros_like(one["vector"]),
              concat=torch.zeros_like(one["concat"]), control_hint=one["control_hint"])
    kw = dict(num_video_frames=T, image_only_indicator=one["image_only_indicator"])
    fn = lambda x, sigma, cc: eng.denoise(x, sigma, cc, **kw)
    with torch.no_grad(), cnet.hint_cache():
        xs = sampler(fn, one["x"].clone(), c, uc=uc)
    torch.cuda.synchronize()
    assert xs.dtype == torch.float32 and torch.isfinite(xs).all()
    e_max, e_rms = _err(xs[:, :, ::2, ::2], G["sample_final_f32"])
    r_max, r_rms = (float(v) for v in G[budget + "sample_final"])
    H.report(f"two sampling steps at full size in {dt}: error (max, rms) = ({e_max:.2e}, {e_rms:.2e}), the reference's own autocast loop ({r_max:.2e}, {r_rms:.2e})")
    assert e_max <= FULL_SIZE_BAR * r_max and e_rms <= FULL_SIZE_BAR * r_rms, (e_max, r_max, e_rms, r_rms)


def test_full_size_networks_in_fp32_meet_the_north_star_tolerance(golden_dir, strict, full_nets):
    """north_star's bar itself — "within 1e-4 rel on ... UNet activations" — at the REAL size of configs[3] on the GPU (round 6;
    the 16-bit runs above are held to the reference's own autocast error, ~1e-2, under which a mis-wired block of small effect
    could hide): ControlNet + ControlledVideoUNet in fp32 on the HIP path against the fp32 tensors of tests/golden/sgm_full.npz
    (ONE evaluation of the imported reference on the CPU, models/csvd.py:34-115, :434-498) — the network output, the last control
    residual and the four intermediate block outputs, each to 1e-4 of its own scale in max norm (and in rms).
    What runs in fp32 (read from the op profile and asserted): this library's GroupNorm(+SiLU) kernels (NCHW, frames and token
    forms), LayerNorm / residual / blend / concat passes, the fp32-math attention kernels (row-tile at S = 9216 / 2304, the
    temporal kernel) — and the vendor libraries for every GEMM and convolution (the MFMA kernels of this library are 16-bit by
    construction and their dtype gates decline fp32; that is a gate, not a fallback: strict mode stays on)."""
    from multiview_inpaint_amd.svd import bench_svd, hip_ops
    from multiview_inpaint_amd.svd import layers as LY
    G = np.load(os.path.join(golden_dir, "sgm_full.npz"))
    torch.backends.cudnn.benchmark = False
    bench_svd.use_shipped_miopen_db()
    old_tf32 = torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = torch.backends.cudnn.allow_tf32 = False
    cunet, cnet = full_nets.get(torch.float32)
    T = H.FULL_T
    inp = H.seeded_inputs(73, T=T, hw=H.FULL_HW, cfg=H.FULL_UNET)
    inp["image_only_indicator"][0, 1] = 1.0
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    kw = dict(num_video_frames=T, image_only_indicator=inp["image_only_indicator"])
    xin = torch.cat([inp["x"], inp["concat"]], 1)
    tt = 0.25 * inp["sigma"].log()
    probes, handles = {}, []
    for name in H.FULL_PROBES:
        handles.append(cunet.get_submodule(name).register_forward_hook(
            lambda m, i, o, name=name: probes.__setitem__(name, LY.to_planes(o).detach().float()[H.FULL_SUB].contiguous())))
    hip_ops.PROFILE = []
    try:
        with torch.no_grad():
            ctrls = cnet(xin, inp["control_hint"], tt, inp["crossattn"], inp["vector"], **kw)
            yc = cunet(xin, tt, inp["crossattn"], inp["vector"], control=list(ctrls), **kw)
        torch.cuda.synchronize()
        kinds = [rec[0] for rec in hip_ops.PROFILE]
    finally:
        hip_ops.PROFILE = None
        torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32 = old_tf32
        for hd in handles:
            hd.remove()
    assert yc.dtype == torch.float32 and torch.isfinite(yc).all() and len(ctrls) == int(G["n_ctrl"])
    count = lambda k: sum(1 for x in kinds if x == k)
    assert count("attention_mfma") == 0 and count("conv3x3_n320") == 0 and count("ff_geglu") == 0, sorted(set(kinds))
    assert count("attention_rowtile") > 0 and count("attention_temporal") > 0 and count("groupnorm") + count("groupnorm_tokens") > 0, sorted(set(kinds))
    worst = 0.0
    for name, got in [("cunet_out", yc), ("ctrl_last", ctrls[-1][H.FULL_SUB])] + [("probe_" + k, probes[k]) for k in H.FULL_PROBES]:
        e_max, e_rms = _err(got, G[name + "_f32"])
        worst = max(worst, e_max, e_rms)
        assert e_max <= 1e-4 and e_rms <= 1e-4, (name, e_max, e_rms)
    H.report(f"full-size step in fp32 on the GPU vs the reference's fp32 CPU evaluation: worst relative error over 6 tensors = {worst:.2e} "
             f"(bar 1e-4); kernels of this library that ran: {sorted(set(kinds))}")


def _attn_ref_chunked(q, k, v, heads, rows=1536):
    """fp64 softmax(QK^T d^-1/2)V, one head and a block of query rows at a time (S = 9216 does not fit otherwise)."""
    B, Sq, HD = q.shape
    D = HD // heads
    out = torch.empty(B, Sq, HD, dtype=torch.float64)
    for b in range(B):
        for h in range(heads):
            kh, vh = k[b, :, h * D:(h + 1) * D].double(), v[b, :, h * D:(h + 1) * D].double()
            for r0 in range(0, Sq, rows):
                s = q[b, r0:r0 + rows, h * D:(h + 1) * D].double() @ kh.T * D ** -0.5
                out[b, r0:r0 + rows, h * D:(h + 1) * D] = torch.softmax(s, -1) @ vh
    return out


def test_attention_mfma_headline_shape_with_peaked_rows(ops):
    """(B, H, S, S, D) = (1, 5, 9216, 9216, 64): the level-0 shape of the 14 x 576x1024 step, 144 KV
    tiles. Rows are peaked on purpose — in every 64-key tile one key per query block is made dominant, with the
    dominance GROWING along the key axis, so the reference exponent of the deferred rescale (threshold 2^8) is forced to
    move many times per row, and some rows have their maximum in the last tile. Also the same through the packed-QKV
    entry (token stride 3*H*D). fp64 reference on the bf16-rounded inputs; bound = bf16 rounding of the output and of P."""
    B, Hh, S, D = 1, 5, 9216, 64
    g = torch.Generator().manual_seed(9216)
    q, k, v = (torch.randn(B, S, Hh * D, generator=g) for _ in range(3))
    # peaked rows. u: one unit direction per head. Queries of block A lean on u; every third 64-key tile t holds ONE key
    # c_t * u with c_t growing along the key axis, so for those rows the running maximum climbs from ~6 to ~40 (natural
    # log units; 9 ... 58 in the kernel's log2 units) in ~48 steps — far more than the 2^8 the deferred rescale lets pass.
    # Block C has its only peak in the very last tile; block B is 32 uniformly peaky rows (one wave's worth).
    n_tiles = S // 64
    u = torch.randn(Hh, D, generator=g)
    u = (u / u.norm(dim=-1, keepdim=True)).reshape(Hh * D)
    q[0, 1000:1096] += 4.0 * u
    for t in range(0, n_tiles, 3):
        k[0, 64 * t + (t % 64)] = u * (12.0 + 68.0 * t / n_tiles)          # logit ~ 4 c_t / sqrt(64) = c_t / 2
    u2 = torch.randn(Hh, D, generator=g)
    u2 = (u2 / u2.norm(dim=-1, keepdim=True)).reshape(Hh * D)
    q[0, S - 40:] += 4.0 * u2
    k[0, S - 3] = u2 * 60.0
    q[0, 5000:5032] *= 5.0
    qb, kb, vb = (t.to(torch.bfloat16) for t in (q, k, v))
    ref = _attn_ref_chunked(qb, kb, vb, Hh)
    assert ops.attention_kernel_kind(S, S, D, torch.bfloat16) == 1
    out = ops.attention(qb.cuda(), kb.cuda(), vb.cuda(), Hh)
    qkv = torch.cat([qb, kb, vb], -1).cuda().contiguous()
    out_p = ops.attention_packed(qkv, Hh)
    torch.cuda.synchronize()
    assert torch.equal(out, out_p)                                           # same arithmetic through the strided entry
    e = (out.double().cpu() - ref).abs()
    scale = ref.abs().amax(dim=(1, 2), keepdim=True)
    assert float((e / scale).max()) < 2e-2
    # rows: relative to the row's own magnitude (a wrong rescale leaves whole rows off by a factor)
    row_err = e.amax(-1) / (ref.abs().amax(-1) + 1e-3)
    assert float(row_err.max()) < 4e-2, float(row_err.max())
    assert float((e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())) < 6e-3


def test_full_size_svd_step_properties(strict):
    """BASELINE.json configs[3] at its real size on the HIP path (14 frames, 576x1024 -> latent 72x128, CFG batch 28,
    bf16 weights, ControlNet + ControlledVideoUNet through Denoiser.forward), checked through size-independent
    properties — there is no reference output at this size (no weights, and the fp32 reference needs hours on the CPU):
      * finite output; every op stayed on the HIP path (strict mode) and the MFMA attention kernel ran at S = 9216;
      * conditioning of c equal to uc => the two CFG halves of the output are equal;
      * image_only_indicator = 1 => frames are independent: permuting the frames permutes the output;
      * the per-sample hint-stem cache does not change the result."""
    from multiview_inpaint_amd.svd import bench_svd, hip_ops
    from multiview_inpaint_amd.svd import layers as LY
    dev = torch.device("cuda")
    T, h, w = 14, 72, 128
    torch.backends.cudnn.benchmark = False                 # immediate-mode solvers: no minute-long search inside a test
    bench_svd.use_shipped_miopen_db()
    eng = bench_svd.build(dev, with_control=True, dtype=torch.bfloat16)
    x, cond, ind = bench_svd.inputs(dev, T, h, w)
    cond = {k: v.bfloat16() for k, v in cond.items()}
    # both CFG halves get the SAME latent and conditioning
    half = lambda t: torch.cat([t[:T], t[:T]], 0)
    x = half(x)
    cond = {k: half(v) for k, v in cond.items()}
    sig = torch.full((2 * T,), 3.0, device=dev)
    kw = dict(num_video_frames=T, image_only_indicator=ind)
    tol = 3e-2                                              # bf16 activations; library GEMM/conv kernels are not bit-reproducible across batch positions

    def rel_(a, b):
        return float((a.float() - b.float()).abs().max() / b.float().abs().max())
    hip_ops.PROFILE = []
    try:
        with torch.no_grad():
            out = eng.denoise(x, sig, cond, **kw)
        torch.cuda.synchronize()
        prof = list(hip_ops.PROFILE)
    finally:
        hip_ops.PROFILE = None
    assert out.shape == (2 * T, 4, h, w) and torch.isfinite(out).all()
    big = [wk for kind, _, _, wk in prof if kind == "attention_mfma"]
    assert len(big) >= 20 and max(big) == 4.0 * 28 * 5 * 9216 * 9216 * 64, (len(big), max(big) if big else 0)
    assert not [k for k, *_ in prof if k == "attention_rowtile"]
    assert rel_(out[:T], out[T:]) < tol
    with torch.no_grad():
        # frames independent under image_only_indicator = 1
        ind1 = torch.ones_like(ind)
        perm = torch.randperm(T, device=dev, generator=torch.Generator(dev).manual_seed(1))
        perm2 = torch.cat([perm, perm + T])
        a = eng.denoise(x, sig, cond, num_video_frames=T, image_only_indicator=ind1)
        b = eng.denoise(x[perm2], sig, {k: v[perm2] for k, v in cond.items()}, num_video_frames=T, image_only_indicator=ind1)
        assert torch.isfinite(a).all() and rel_(b, a[perm2]) < tol
        assert rel_(a, out) > tol                         # the temporal path does contribute: switching it off moves the output
        # hint-stem cache on == off
        with eng.control_model.hint_cache():
            c1 = eng.denoise(x, sig, cond, **kw)
            c2 = eng.denoise(x, sig * 0.5, cond, **kw)
        d2 = eng.denoise(x, sig * 0.5, cond, **kw)
    assert rel_(c1, out) < tol and rel_(c2, d2) < tol


def test_controlnet_on_a_side_stream_gives_the_same_step():
    """SVDInpaintEngine.apply_model runs the ControlNet on a side stream beside the UNet's encoder (engine.TWO_STREAMS) and
    joins it where the first residual is popped: same kernels, same inputs — the step must equal the one-stream step (fp32
    small networks: bit for bit up to the vendor GEMMs' own run-to-run noise, bound 1e-6), repeatedly (the streams' buffers
    and caches are reused), and with the hint-stem cache of sample()."""
    from multiview_inpaint_amd.svd import engine as E
    from multiview_inpaint_amd.svd.schedule import Denoiser
    from multiview_inpaint_amd.svd.unet import ControlNet, ControlledVideoUNet
    cunet = ControlledVideoUNet(**H.SMALL_UNET).eval()
    cunet.load_state_dict(H.seeded_state_dict(cunet, 11))
    cnet = ControlNet(**H.SMALL_CTRL).eval()
    sd = H.seeded_state_dict(cnet, 12)
    for k in sd:                                                     # zero convolutions would hide the ControlNet entirely
        if "zero_convs" in k or "middle_block_out" in k:
            sd[k] = torch.randn(sd[k].shape, generator=torch.Generator().manual_seed(len(k))) * 0.05
    cnet.load_state_dict(sd)
    den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    eng = E.SVDInpaintEngine(cunet, cnet, den).cuda()
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in H.seeded_inputs(21).items()}
    cond = {"concat": inp["concat"], "crossattn": inp["crossattn"], "vector": inp["vector"], "control_hint": inp["control_hint"]}
    kw = dict(num_video_frames=H.T_FRAMES, image_only_indicator=inp["image_only_indicator"])
    outs = {}
    old = E.TWO_STREAMS
    try:
        with torch.no_grad():
            for mode in (False, True, True, False):
                E.TWO_STREAMS = mode
                outs.setdefault(mode, []).append(eng.denoise(inp["x"], inp["sigma"], cond, **kw))
            E.TWO_STREAMS = True
            with cnet.hint_cache():
                cached = [eng.denoise(inp["x"], inp["sigma"], cond, **kw) for _ in range(2)]
        torch.cuda.synchronize()
    finally:
        E.TWO_STREAMS = old
    ref = outs[False][0].double()
    assert float(ref.abs().max()) > 0 and rel(outs[False][1], ref) < 1e-6
    for o in outs[True] + cached:
        assert rel(o, ref) < 1e-6



@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
def test_groupnorm_token_major_in_and_out(ops, dtype, tol):
    """GroupNorm(+SiLU) with tokens on both sides (the norm between two channels-last convolutions): against fp64 GroupNorm of the
    transposed tensor, with the fused timestep bias, a group whose mean is 30 standard deviations away, token counts that are not
    multiples of the chunk (64 / 48 / 24 rows), 320 ... 2560 channels."""
    g = torch.Generator().manual_seed(51)
    for N, S, C, G, with_bias, silu in [(3, 1000, 320, 32, True, True), (2, 577, 640, 32, False, True), (2, 144, 1280, 32, True, False),
                                        (1, 40, 2560, 32, True, True), (5, 64, 64, 8, False, True)]:
        t = (torch.randn(N, S, C, generator=g) * 1.4 + 0.2).to(dtype)
        t[0, :, :5] += 40.0
        w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
        cb = torch.randn(N, C, generator=g) if with_bias else None
        xf = t.double().transpose(1, 2) + (0 if cb is None else cb.double()[:, :, None])
        ref = F.group_norm(xf, G, w.double(), b.double(), 1e-5)
        ref = (F.silu(ref) if silu else ref).transpose(1, 2)
        y = ops.group_norm_silu_tok2tok(t.cuda(), G, w.cuda(), b.cuda(), 1e-5, silu, chan_bias=None if cb is None else cb.cuda())
        assert y.shape == t.shape and y.dtype == dtype and rel(y, ref) < tol, (N, S, C)
    with pytest.raises(ValueError):
        ops.group_norm_silu_tok2tok(torch.zeros(1, 8, 36, device="cuda", dtype=torch.bfloat16), 6, torch.ones(36).cuda(),
                                    torch.zeros(36).cuda(), 1e-5, False)                         # 36 channels: not a multiple of the vector


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 3.0 / 128), (torch.float16, 3.0 / 1024)])
def test_resblock_with_channels_last_convolutions_equals_the_nchw_path(dtype, tol):
    """ResBlock._forward_fused in reduced precision hands its two 3x3 convolutions channels-last tensors (tokens from the first norm,
    token-major norm in between, tokens into the last add): same block, same input through both routes (layers.NHWC_CONVS),
    identity skip and 1x1 skip convolution, against the fp32 NCHW evaluation of the same weights."""
    from multiview_inpaint_amd.svd import layers as LY
    g = torch.Generator().manual_seed(8)
    for cin, cout, hw in [(64, 64, (24, 40)), (128, 64, (16, 24))]:
        blk = LY.ResBlock(cin, 256, 0.0, out_channels=cout, dims=2).eval()
        with torch.no_grad():
            for p in blk.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.3))
        x = torch.randn(4, cin, *hw, generator=g)
        emb = torch.randn(4, 256, generator=g)
        with torch.no_grad():
            ref = blk.double()(x.double(), emb.double())
            blk = blk.to(dtype).cuda()
            xs, es = x.to(dtype).cuda(), emb.to(dtype).cuda()
            outs = {}
            old = LY.NHWC_CONVS
            try:
                for mode in (False, True):
                    LY.NHWC_CONVS = mode
                    outs[mode] = blk(xs, es)
            finally:
                LY.NHWC_CONVS = old
        assert outs[True].shape == ref.shape and outs[True].is_contiguous()
        assert rel(outs[True], ref) < tol and rel(outs[False], ref) < tol
        assert rel(outs[True], outs[False].double()) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
@pytest.mark.parametrize("N,H,W,C", [(2, 24, 32, 320), (3, 9, 16, 64), (1, 7, 5, 128), (5, 1, 1, 64), (2, 3, 96, 640), (1, 40, 33, 192)])
def test_conv3x3_n320_equals_conv2d(dtype, tol, N, H, W, C):
    """csrc/linear_n320.hip as an implicit GEMM: 3x3 / padding 1 convolution to 320 channels on token-major activations against
    F.conv2d in fp64 on the same rounded inputs (openaimodel.py:256-275 `in_layers[2]`): image borders, rows that straddle image
    rows and images inside one wave (W = 5, 16, 33), a last partial row block, one-pixel images, bias on and off."""
    from multiview_inpaint_amd.svd import hip_ops
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + W)
    x = torch.randn(N, C, H, W, generator=g).to(dtype)
    w = (torch.randn(320, C, 3, 3, generator=g) * (1.0 / (9 * C) ** 0.5)).to(dtype)
    b = torch.randn(320, generator=g)
    assert hip_ops.conv3x3_n320_supported(C, 320, dtype)
    tok = x.permute(0, 2, 3, 1).reshape(N, H * W, C).contiguous().cuda()
    wt = hip_ops.conv3x3_n320_weight(w.cuda())
    for bias, split in ((None, False), (b, False), (b, True)):     # split: K over several blocks per tile + the fp32 reduction (small images)
        ref = F.conv2d(x.double(), w.double(), None if bias is None else bias.double(), padding=1)
        out = hip_ops.conv3x3_n320(tok, wt, None if bias is None else bias.cuda(), H, W, split=split)
        torch.cuda.synchronize()
        assert out.shape == (N, H * W, 320)
        got = out.view(N, H, W, 320).permute(0, 3, 1, 2).double().cpu()
        assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
        assert rel(got, ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
@pytest.mark.parametrize("N,H,W,C,Co", [(2, 12, 16, 128, 640), (1, 9, 16, 64, 1280), (3, 5, 7, 192, 960)])
def test_conv3x3_n320_column_groups(dtype, tol, N, H, W, C, Co):
    """C_out = 640 / 960 / 1280: one launch, a block per (256 rows, 320 output channels); bias per group."""
    from multiview_inpaint_amd.svd import hip_ops
    g = torch.Generator().manual_seed(N + H + Co)
    x = torch.randn(N, C, H, W, generator=g).to(dtype)
    w = (torch.randn(Co, C, 3, 3, generator=g) * (1.0 / (9 * C) ** 0.5)).to(dtype)
    b = torch.randn(Co, generator=g)
    tok = x.permute(0, 2, 3, 1).reshape(N, H * W, C).contiguous().cuda()
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    for split in (False, True):
        out = hip_ops.conv3x3_n320(tok, hip_ops.conv3x3_n320_weight(w.cuda()), b.cuda(), H, W, split=split)
        got = out.view(N, H, W, Co).permute(0, 3, 1, 2).double().cpu()
        assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
@pytest.mark.parametrize("B,T,S,C,Co", [(2, 14, 40, 320, 320), (1, 3, 33, 64, 640), (3, 1, 8, 128, 320), (2, 2, 300, 64, 320), (1, 14, 5, 192, 320)])
def test_conv3t_n320_equals_conv3d(dtype, tol, B, T, S, C, Co):
    """The (3, 1, 1) frame-axis convolution of VideoResBlock.time_stack (video_model.py:41-54, openaimodel.py:256-275 with dims = 3)
    on token-major frames [(b T), S, C] against F.conv3d in fp64 on b c t h w: first / last frame of every video (zero padding,
    nothing leaks between videos), T = 1, frames smaller and larger than a wave's 32 rows."""
    from multiview_inpaint_amd.svd import hip_ops
    g = torch.Generator().manual_seed(B * 100 + T * 10 + S)
    tok = torch.randn(B * T, S, C, generator=g).to(dtype)
    w = (torch.randn(Co, C, 3, 1, 1, generator=g) * (1.0 / (3 * C) ** 0.5)).to(dtype)
    b = torch.randn(Co, generator=g)
    x5 = tok.double().view(B, T, S, 1, C).permute(0, 4, 1, 2, 3)                           # b c t h w (w = 1)
    ref = F.conv3d(x5, w.double(), b.double(), 1, (1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(B * T, S, Co)
    for split in (False, True):
        out = hip_ops.conv3t_n320(tok.cuda(), hip_ops.conv3t_n320_weight(w.cuda()), b.cuda(), T, split=split)
        torch.cuda.synchronize()
        assert out.shape == (B * T, S, Co)
        assert (out.double().cpu() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
@pytest.mark.parametrize("N,H,W,C,Co", [(2, 24, 32, 320, 320), (1, 9, 15, 64, 640), (3, 8, 6, 128, 320), (1, 1, 1, 64, 320), (2, 7, 40, 192, 320)])
def test_conv3x3_n320_stride_2(dtype, tol, N, H, W, C, Co):
    """Downsample.op (openaimodel.py:150-166: 3x3, stride 2, padding 1) in the implicit-GEMM kernel: a row = an output pixel, its taps
    around input pixel (2 yo, 2 xo); even and odd image sizes (the last row / column of taps falls outside only for odd sizes), with
    and without the K split, against F.conv2d in fp64."""
    from multiview_inpaint_amd.svd import hip_ops
    g = torch.Generator().manual_seed(N * 100 + H + W)
    x = torch.randn(N, C, H, W, generator=g).to(dtype)
    w = (torch.randn(Co, C, 3, 3, generator=g) * (1.0 / (9 * C) ** 0.5)).to(dtype)
    b = torch.randn(Co, generator=g)
    tok = x.permute(0, 2, 3, 1).reshape(N, H * W, C).contiguous().cuda()
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    for split in (False, True):
        out = hip_ops.conv3x3_n320(tok, hip_ops.conv3x3_n320_weight(w.cuda()), b.cuda(), H, W, stride=2, split=split)
        torch.cuda.synchronize()
        assert out.shape == (N, Ho * Wo, Co)
        got = out.view(N, Ho, Wo, Co).permute(0, 3, 1, 2).double().cpu()
        assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 1.0 / 1024)])
@pytest.mark.parametrize("N,H,W,C,Co", [(2, 12, 16, 320, 320), (1, 9, 15, 64, 640), (3, 4, 3, 128, 320), (1, 1, 1, 64, 320), (2, 7, 20, 192, 320),
                                        (28, 36, 64, 640, 640)])
def test_conv3x3_up2_n320_equals_conv2d_of_the_upsampled_image(dtype, tol, N, H, W, C, Co):
    """Upsample.conv (openaimodel.py:107-150: F.interpolate(scale_factor=2, mode="nearest") then 3x3 / padding 1) with the upsampling in
    the implicit-GEMM kernel's addressing (mvi_conv3x3_up2_n320, kUps): a row = a pixel (y, x) of the 2 H x 2 W image, its taps read
    source pixel ((y + dy) >> 1, (x + dx) >> 1) of the H x W tokens; even and odd source sizes, one pixel, row blocks that straddle
    images — against F.conv2d of the upsampled image in fp64 (the last case, the level 1 -> 0 shape of the 576x1024 step, against the
    kernel's own plain form on the materialised upsampled tokens: bit for bit)."""
    from multiview_inpaint_amd.svd import hip_ops
    g = torch.Generator().manual_seed(N * 100 + H + W)
    x = torch.randn(N, C, H, W, generator=g).to(dtype)
    w = (torch.randn(Co, C, 3, 3, generator=g) * (1.0 / (9 * C) ** 0.5)).to(dtype)
    b = torch.randn(Co, generator=g)
    tok = x.permute(0, 2, 3, 1).reshape(N, H * W, C).contiguous().cuda()
    wt = hip_ops.conv3x3_n320_weight(w.cuda())
    out = hip_ops.conv3x3_n320(tok, wt, b.cuda(), H, W, up2=True)
    torch.cuda.synchronize()
    assert out.shape == (N, 4 * H * W, Co)
    up = tok.view(N, H, 1, W, 1, C).expand(N, H, 2, W, 2, C).reshape(N, 4 * H * W, C)
    assert torch.equal(out, hip_ops.conv3x3_n320(up, wt, b.cuda(), 2 * H, 2 * W, split=False))     # same products, same order
    if N * H * W <= 4096:
        ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
        got = out.view(N, 2 * H, 2 * W, Co).permute(0, 3, 1, 2).double().cpu()
        assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


def test_downsample_on_tokens_equals_the_library_route():
    from multiview_inpaint_amd.svd import hip_ops
    from multiview_inpaint_amd.svd import layers as LY
    g = torch.Generator().manual_seed(4)
    ds = LY.Downsample(320, True, dims=2, out_channels=320).eval()
    with torch.no_grad():
        ds.op.weight.copy_(torch.randn(ds.op.weight.shape, generator=g) * 0.02)
        ds.op.bias.copy_(torch.randn(320, generator=g) * 0.3)
    x = torch.randn(2, 320, 16, 24, generator=g).bfloat16()
    ds = ds.bfloat16()
    ref = F.conv2d(x.double(), ds.op.weight.detach().double(), ds.op.bias.detach().double(), stride=2, padding=1)   # (the rounded parameters)
    ds = ds.cuda()
    old = LY.CONV_N320, LY.CONV_N320_MIN_BLOCKS
    try:
        LY.CONV_N320_MIN_BLOCKS = 1
        for mode in (False, True):
            LY.CONV_N320 = mode
            hip_ops.PROFILE = []
            with torch.no_grad():
                y = ds(x.cuda())
            torch.cuda.synchronize()
            assert sum(1 for rec in hip_ops.PROFILE if rec[0] == "conv3x3_n320") == (1 if mode else 0)
            err = (y.double().cpu() - ref).abs().max().item()
            # (the library's stride-2 solver is the less accurate of the two routes: 0.043 against 0.012 on this input)
            assert y.shape == ref.shape and y.is_contiguous() and err <= ref.abs().max().item() * (1 if mode else 2) / 128, (mode, err)
    finally:
        LY.CONV_N320, LY.CONV_N320_MIN_BLOCKS = old
        hip_ops.PROFILE = None


def test_conv3x3_n320_k_split_policy():
    """Which shapes split K (mvi_conv3x3_n320_workspace_bytes > 0): the level-3 images of the SVD step do, levels 0-2 do not, and a K
    too short for 8 chunks per part does not either."""
    from multiview_inpaint_amd import _lib
    L = _lib.lib()
    assert L.mvi_conv3x3_n320_workspace_bytes(28, 9, 16, 1280, 1280, 1) == 4 * 4096 * 1280 * 4   # 64 blocks -> 4 parts of 45 chunks
    assert L.mvi_conv3x3_n320_workspace_bytes(28, 9, 16, 2560, 1280, 1) == 4 * 4096 * 1280 * 4
    assert L.mvi_conv3x3_n320_workspace_bytes(28, 18, 32, 1280, 1280, 1) == 0 and L.mvi_conv3x3_n320_workspace_bytes(28, 72, 128, 320, 320, 1) == 0
    assert L.mvi_conv3x3_n320_workspace_bytes(28, 18, 32, 1280, 1280, 2) == 4 * 4096 * 1280 * 4   # Downsample at level 2: 9x16 outputs
    assert L.mvi_conv3x3_n320_workspace_bytes(1, 4, 4, 64, 320, 1) == 0                             # 9 chunks: nothing to split
    assert L.mvi_conv3t_n320_workspace_bytes(2, 14, 144, 1280, 1280) == 4 * 4096 * 1280 * 4       # 60 chunks


def test_conv3x3_n320_refuses_other_shapes():
    from multiview_inpaint_amd.svd import hip_ops
    assert not hip_ops.conv3x3_n320_supported(320, 4, torch.bfloat16)          # the output convolution (320 -> 4): the library's
    assert not hip_ops.conv3x3_n320_supported(320, 480, torch.bfloat16)        # not whole groups of 320 outputs
    assert not hip_ops.conv3x3_n320_supported(8, 320, torch.bfloat16)          # the stem's 8 channels: csrc/stem_conv.hip
    assert not hip_ops.conv3x3_n320_supported(320, 320, torch.float32)
    tok = torch.zeros(1, 12, 64, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        hip_ops.conv3x3_n320(tok, torch.zeros(320, 9 * 64, device="cuda", dtype=torch.bfloat16), None, 3, 5)   # H W != 12


def test_resblock_at_320_channels_takes_the_implicit_gemm_convolution():
    """At 320 output channels ResBlock._forward_fused's convolutions run in csrc/linear_n320.hip (layers.CONV_N320): same block through
    both routes and against the fp64 NCHW evaluation; the op profile shows which kernel ran."""
    from multiview_inpaint_amd.svd import hip_ops
    from multiview_inpaint_amd.svd import layers as LY
    g = torch.Generator().manual_seed(18)
    for cin in (320, 640):
        blk = LY.ResBlock(cin, 256, 0.0, out_channels=320, dims=2).eval()
        with torch.no_grad():
            for p in blk.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * (0.03 if p.dim() > 1 else 0.3))
        x = torch.randn(2, cin, 16, 24, generator=g)
        emb = torch.randn(2, 256, generator=g)
        with torch.no_grad():
            ref = blk.double()(x.double(), emb.double())
            blk = blk.to(torch.bfloat16).cuda()
            xs, es = x.bfloat16().cuda(), emb.bfloat16().cuda()
            outs = {}
            old = LY.CONV_N320, LY.CONV_N320_MIN_BLOCKS
            try:
                LY.CONV_N320_MIN_BLOCKS = 1                  # (a test-sized image is far below the launch size the gate asks for)
                for mode in (False, True):
                    LY.CONV_N320 = mode
                    hip_ops.PROFILE = []
                    outs[mode] = blk(xs, es)
                    torch.cuda.synchronize()
                    assert sum(1 for rec in hip_ops.PROFILE if rec[0] == "conv3x3_n320") == (2 if mode else 0)
            finally:
                LY.CONV_N320, LY.CONV_N320_MIN_BLOCKS = old
                hip_ops.PROFILE = None
        assert rel(outs[True], ref) < 3.0 / 128 and rel(outs[True], outs[False].double()) < 3.0 / 128


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,C,H,W", [(2, 64, 8, 16), (1, 72, 5, 8), (3, 320, 9, 16), (1, 8, 24, 40)])
def test_planes_to_tokens_and_back(dtype, N, C, H, W):
    """mvi_planes_to_tokens: "b c h w -> b (h w) c" alone and with Upsample's nearest 2x folded in (openaimodel.py:118-134), bit for bit
    against permute / F.interpolate; mvi_tokens_to_planes_add without x_in is its inverse (+ bias)."""
    from multiview_inpaint_amd.svd import hip_ops
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g).to(dtype).cuda()
    t1 = hip_ops.planes_to_tokens(x)
    assert torch.equal(t1, x.permute(0, 2, 3, 1).reshape(N, H * W, C))
    t2 = hip_ops.planes_to_tokens(x, upsample=2)
    up = F.interpolate(x.float(), scale_factor=2, mode="nearest").to(dtype)
    assert torch.equal(t2, up.permute(0, 2, 3, 1).reshape(N, 4 * H * W, C))
    assert torch.equal(hip_ops.tokens_to_planes_add(t2, None, spatial=(2 * H, 2 * W)), up)
    b = torch.randn(C, generator=g).cuda()
    back = hip_ops.tokens_to_planes_add(t1, None, b, spatial=(H, W))
    assert torch.equal(back, (x.float() + b.view(1, C, 1, 1)).to(dtype))
    with pytest.raises(ValueError):
        hip_ops.tokens_to_planes_add(t1, None, spatial=(H, W + 1))


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 128), (torch.float16, 2.0 / 1024)])   # (the library adds the bias in a second rounding)
def test_upsample_on_tokens_equals_interpolate_then_conv(dtype, tol):
    """Upsample.forward (openaimodel.py:118-134) through planes_to_tokens(upsample = 2) + the implicit-GEMM convolution + the layout
    change back, against F.interpolate + F.conv2d in fp64; both routes of the module (layers.CONV_N320)."""
    from multiview_inpaint_amd.svd import hip_ops
    from multiview_inpaint_amd.svd import layers as LY
    g = torch.Generator().manual_seed(3)
    up = LY.Upsample(640, True, dims=2, out_channels=640).eval()
    with torch.no_grad():
        up.conv.weight.copy_(torch.randn(up.conv.weight.shape, generator=g) * 0.02)
        up.conv.bias.copy_(torch.randn(640, generator=g) * 0.3)
    x = torch.randn(2, 640, 6, 8, generator=g).to(dtype)
    up = up.to(dtype).cuda()
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), up.conv.weight.double().cpu(), up.conv.bias.double().cpu(), padding=1)
    old = LY.CONV_N320, LY.CONV_N320_MIN_BLOCKS
    outs = {}
    try:
        LY.CONV_N320_MIN_BLOCKS = 1
        for mode in (False, True):
            LY.CONV_N320 = mode
            hip_ops.PROFILE = []
            with torch.no_grad():
                outs[mode] = up(x.cuda())
            torch.cuda.synchronize()
            assert sum(1 for rec in hip_ops.PROFILE if rec[0] == "conv3x3_n320") == (1 if mode else 0)
    finally:
        LY.CONV_N320, LY.CONV_N320_MIN_BLOCKS = old
        hip_ops.PROFILE = None
    for o in outs.values():
        assert o.shape == ref.shape and o.is_contiguous() and (o.double().cpu() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.0 / 64), (torch.float16, 1.0 / 512)])
@pytest.mark.parametrize("b,T,S,C", [(2, 14, 144, 320), (1, 3, 40, 64), (3, 1, 17, 96)])
def test_groupnorm_tok2tok_with_temporal_statistics(ops, dtype, tol, b, T, S, C):
    """mvi_groupnorm_silu_tok2tok_frames: token-major [(b T), S, C] with a group's statistics over all T frames of a video and a
    per-frame chan_bias — the GroupNorm of VideoResBlock.time_stack on b c t h w (video_model.py:71-75, openaimodel.py:341-352)."""
    g = torch.Generator().manual_seed(C + T)
    x = (torch.randn(b * T, S, C, generator=g) * 1.5 + 0.2).to(dtype)
    w, bb = torch.randn(C, generator=g), torch.randn(C, generator=g)
    cb = torch.randn(b * T, C, generator=g)
    for chan_bias in (None, cb):
        xf = x.double() if chan_bias is None else x.double() + chan_bias.double()[:, None, :]
        x5 = xf.reshape(b, T * S, C).transpose(1, 2)                                                   # b c (t s)
        ref = F.silu(F.group_norm(x5, 32, w.double(), bb.double(), 1e-5)).transpose(1, 2).reshape(b * T, S, C)
        y = ops.group_norm_silu_tok2tok(x.cuda(), 32, w.cuda(), bb.cuda(), 1e-5, True, chan_bias=None if chan_bias is None else chan_bias.cuda(),
                                        frames=T)
        assert y.dtype == dtype and (y.double().cpu() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    with pytest.raises(ValueError):
        ops.group_norm_silu_tok2tok(x.cuda(), 32, w.cuda(), bb.cuda(), 1e-5, True, frames=T + 1)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 64), (torch.float16, 1.0 / 512)])
@pytest.mark.parametrize("taps,N,H,W,C,Co,frames", [(9, 8, 64, 64, 320, 320, 1), (9, 4, 64, 64, 640, 640, 1), (9, 2, 128, 128, 1280, 1280, 1),
                                                    (3, 4, 4, 8192, 320, 320, 4), (3, 2, 4, 8192, 640, 640, 4)])
def test_groupnorm_statistics_from_the_convolution_epilogue(ops, dtype, tol, taps, N, H, W, C, Co, frames):
    """mvi_conv3x3_n320_gnstats / mvi_conv3t_n320_gnstats + mvi_groupnorm_silu_tok2tok_pre (round 5): the convolution that feeds a
    ResBlock's second GroupNorm (openaimodel.py:292-305, :339-343; the temporal twin video_model.py:41-54) leaves that norm's
    (count, mean, M2) partials behind, and the norm merges them instead of reading the tensor a second time. The convolution's
    output is bit-identical to the plain launch; the norm's output equals the three-launch form's to the type's rounding and meets
    the fp64 GroupNorm of the stored tensor (+ the per-sample channel bias, statistics over `frames` frames for the 3-tap form) at
    the bar of the other token-norm tests. Group widths 10 / 20 / 40 channels (C_out 320 / 640 / 1280), one to four column groups."""
    g = torch.Generator().manual_seed(C + taps)
    if taps == 9:
        samples, S = N, H * W
        tok = (torch.randn(N, S, C, generator=g) * 0.7).to(dtype).cuda()
        wt = ops.conv3x3_n320_weight((torch.randn(Co, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).to(dtype).cuda())
        run = lambda gn: ops.conv3x3_n320(tok, wt, None, H, W, gn=gn)
    else:                                                            # N videos of H frames of W pixels
        samples, S = N * H, W
        tok = (torch.randn(samples, S, C, generator=g) * 0.7).to(dtype).cuda()
        wt = ops.conv3t_n320_weight((torch.randn(Co, C, 3, 1, 1, generator=g) / (1.7 * C ** 0.5)).to(dtype).cuda())
        run = lambda gn: ops.conv3t_n320(tok, wt, None, H, gn=gn)
    assert ops.conv_n320_gnstats_supported(samples * S, taps, C, Co, S, 32)
    w, bb = (1.0 + 0.1 * torch.randn(Co, generator=g)).cuda(), (0.1 * torch.randn(Co, generator=g)).cuda()
    cb = (0.5 * torch.randn(samples, Co, generator=g)).cuda()
    plain = run(None)
    for chan_bias in (cb, None):
        out, stats = run((32, chan_bias))
        assert torch.equal(out, plain) and stats.chunks == S // 256 and stats.part.numel() == samples * (S // 256) * 32 * 3
        y = ops.group_norm_silu_tok2tok(out, 32, w, bb, 1e-5, True, chan_bias=chan_bias, frames=frames, partials=stats)
        y3 = ops.group_norm_silu_tok2tok(out, 32, w, bb, 1e-5, True, chan_bias=chan_bias, frames=frames)
        xf = out.double().cpu() if chan_bias is None else out.double().cpu() + chan_bias.double().cpu()[:, None, :]
        x5 = xf.reshape(samples // frames, frames * S, Co).transpose(1, 2)
        ref = F.silu(F.group_norm(x5, 32, w.double().cpu(), bb.double().cpu(), 1e-5)).transpose(1, 2).reshape(samples, S, Co)
        scale = max(1.0, ref.abs().max().item())
        e_pre, e_3 = (y.double().cpu() - ref).abs().max().item() / scale, (y3.double().cpu() - ref).abs().max().item() / scale
        assert e_pre <= tol and e_pre <= 1.5 * e_3 + 1e-6, (e_pre, e_3)
        # the statistics themselves: merged per (sample, group) they are the fp64 moments of the stored tensor
        part = stats.part.view(samples, S // 256, 32, 3).double().cpu()
        cnt, mean = part[..., 0], part[..., 1]
        gmean = (cnt * mean).sum(1) / cnt.sum(1)
        want = xf.reshape(samples, S, 32, Co // 32).mean(dim=(1, 3))
        assert (gmean - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    with pytest.raises(ValueError):
        ops.group_norm_silu_tok2tok(plain, 16, w, bb, 1e-5, True, partials=stats)      # another norm's groups


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.0 / 64), (torch.float16, 1.0 / 512)])
def test_groupnorm_statistics_from_the_convolution_epilogue_with_a_large_common_mean(ops, dtype, tol):
    """the kStats epilogue forms a block's M2 from RAW fp32 moments of the stored outputs (sum, sum of squares per channel over
    256 rows), which cancels when a group's mean is large against its spread — unlike the shifted sums of the standalone statistics pass.
    The envelope, measured: a convolution whose outputs share a mean of ~1 with a spread of ~0.08 (|mean| / std ~ 12: every weight equal, so
    all channels of a group agree; the spread is the image border's missing taps + noise) — the merged variance stays within 2e-3 of the
    fp64 variance of the stored tensor and the norm's output within the token norms' bar, equal to the three-launch form's to 1.5 x.
    (Raw moments lose ~R^2 2^-24 sqrt(n) of the variance: beyond R ~ 50 in f16 the standalone pass should be used — MVI_SVD_GN_STATS_FROM_CONV=0.)"""
    g = torch.Generator().manual_seed(17)
    N, Hh, Ww, C, Co = 8, 64, 64, 320, 320                            # (128 blocks: the launch is not K-split and may leave statistics)
    assert ops.conv_n320_gnstats_supported(N * Hh * Ww, 9, C, Co, Hh * Ww, 32)
    tok = (1.0 + 0.02 * (9 * C) ** 0.5 * torch.randn(N, Hh * Ww, C, generator=g)).to(dtype).cuda()
    wt = ops.conv3x3_n320_weight(torch.full((Co, C, 3, 3), 1.0 / (9 * C)).to(dtype).cuda())
    w, bb = (1.0 + 0.1 * torch.randn(Co, generator=g)).cuda(), (0.1 * torch.randn(Co, generator=g)).cuda()
    out, stats = ops.conv3x3_n320(tok, wt, None, Hh, Ww, gn=(32, None))
    xf = out.double().cpu()
    R = float(xf.mean().abs() / xf.std())
    assert R > 8, R
    part = stats.part.view(N, Hh * Ww // 256, 32, 3).double().cpu()
    cnt, mean, m2 = part[..., 0], part[..., 1], part[..., 2]
    gmean = (cnt * mean).sum(1) / cnt.sum(1)
    gvar = (m2 + cnt * (mean - gmean[:, None]) ** 2).sum(1) / cnt.sum(1)
    xg = xf.reshape(N, Hh * Ww, 32, Co // 32)
    want_var = xg.var(dim=(1, 3), unbiased=False)
    assert float(((gvar - want_var).abs() / want_var).max()) < 2e-3, float(((gvar - want_var).abs() / want_var).max())
    y = ops.group_norm_silu_tok2tok(out, 32, w, bb, 1e-5, True, partials=stats)
    y3 = ops.group_norm_silu_tok2tok(out, 32, w, bb, 1e-5, True)
    ref = F.silu(F.group_norm(xf.transpose(1, 2), 32, w.double().cpu(), bb.double().cpu(), 1e-5)).transpose(1, 2)
    scale = max(1.0, ref.abs().max().item())
    e_pre, e_3 = (y.double().cpu() - ref).abs().max().item() / scale, (y3.double().cpu() - ref).abs().max().item() / scale
    H.report(f"{dtype}: |mean| / std = {R:.1f}: variance from the convolution's raw moments within {float(((gvar - want_var).abs() / want_var).max()):.1e} of fp64; "
             f"norm output error {e_pre:.2e} (three-launch form {e_3:.2e})")
    assert e_pre <= tol and e_pre <= 1.5 * e_3 + 1e-6, (e_pre, e_3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_token_forms_of_the_resblock_adds(ops, dtype):
    """mvi_planes_add_to_tokens = tokens_to_planes_add with the result left token-major (bit for bit: one rounding of the same fp32
    sum); mvi_tokens_blend_to_planes = bias_residual_blend (AlphaBlender after the temporal skip add) fed with tokens."""
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 4, 72, 6, 12
    x = torch.randn(N, C, H, W, generator=g).to(dtype).cuda()
    t = torch.randn(N, H * W, C, generator=g).to(dtype).cuda()
    bias = torch.randn(C, generator=g).cuda()
    alpha = torch.rand(N, generator=g).cuda()
    for bb in (None, bias):
        planes = ops.tokens_to_planes_add(t, x, bb)
        toks = ops.planes_add_to_tokens(x, t, bb)
        assert torch.equal(toks, planes.permute(0, 2, 3, 1).reshape(N, H * W, C))
    base = torch.randn(N, H * W, C, generator=g).to(dtype).cuda()
    out = ops.tokens_blend_to_planes(t, base, bias, alpha, (H, W))
    to_planes = lambda z: z.view(N, H, W, C).permute(0, 3, 1, 2).contiguous()
    ref = ops.bias_residual_blend(to_planes(t), bias, to_planes(base), alpha)      # (same formula; the compilers contract it differently)
    ulp = {torch.float32: 2.0 ** -22, torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[dtype]
    assert (out.double() - ref.double()).abs().max().item() <= ulp * max(1.0, ref.abs().max().item())
    want = to_planes(base).double() + (1.0 - alpha.double()).view(N, 1, 1, 1) * (to_planes(t).double() + bias.double().view(1, C, 1, 1))
    assert (out.double() - want).abs().max().item() <= ulp * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_row_tails_of_the_token_stream_and_the_statistics_they_leave(ops, dtype):
    """mvi_rows_fused_gnstats (csrc/groupnorm_tokens.hip gt_fused_kernel), the block tails of the token-major residual stream: add
    (a + b + bias: openaimodel.py:354, attention.py:717-722), blend (base + (1 - alpha)(a + bias): video_model.py:67-81 after
    util.py:358-372) and concat ((a | b + base): csvd.py:84-91) against fp64 to one rounding, on shapes whose row count ends inside a
    chunk and inside a set; and the (count, mean, M2) partials they leave: GroupNorm(32) + SiLU from them equals the three-launch
    norm of the same tensor (per sample, and with the statistics of the temporal norm over 3 frames) and fp64."""
    g = torch.Generator().manual_seed(61)
    ulp = {torch.float32: 2.0 ** -22, torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[dtype]
    for (N, S, C) in [(6, 8 * 16, 320), (3, 36 * 64 - 8, 640), (6, 9 * 16, 1280)]:
        mk = lambda *sh: (torch.randn(*sh, generator=g) * 1.5 + 0.7).to(dtype).cuda()
        a, b, base = mk(N, S, C), mk(N, S, C), mk(N, S, C)
        bias, alpha = torch.randn(C, generator=g).cuda(), torch.rand(N, generator=g).cuda()
        w, bb = (torch.randn(C, generator=g) * 0.5 + 1).cuda(), torch.randn(C, generator=g).cuda()
        d = lambda z: z.double().cpu()
        cases = {
            "add": (dict(b=b, bias=bias), d(a) + d(b) + d(bias)),
            "add1": (dict(bias=bias), d(a) + d(bias)),
            "blend": (dict(bias=bias, base=base, alpha=alpha), d(base) + (1.0 - d(alpha)).view(N, 1, 1) * (d(a) + d(bias))),
            "concat": (dict(b=b, base=base, concat=True), torch.cat([d(a), d(b) + d(base)], -1)),
            "concat1": (dict(b=b, concat=True), torch.cat([d(a), d(b)], -1)),
        }
        for name, (kw, want) in cases.items():
            plain, none = ops.rows_fused(a, **kw)
            assert none is None and plain.shape == want.shape
            assert (d(plain) - want).abs().max().item() <= ulp * max(1.0, want.abs().max().item()), name
            if dtype == torch.float32:
                continue                                       # (the norm that takes producer statistics is the bf16 / f16 one)
            Cc = want.shape[-1]
            out, st = ops.rows_fused(a, groups=32, **kw)
            assert torch.equal(out, plain) and st is not None and st.groups == 32 and st.chan_bias is None, name
            wc, bc = (w, bb) if Cc == C else (torch.cat([w, w]), torch.cat([bb, bb]))
            for frames in (1, 3):
                y = ops.group_norm_silu_tok2tok(out, 32, wc, bc, 1e-5, True, frames=frames, partials=st)
                y3 = ops.group_norm_silu_tok2tok(out, 32, wc, bc, 1e-5, True, frames=frames)
                xf = d(out).reshape(N // frames, frames * S, Cc)
                ref = F.silu(F.group_norm(xf.transpose(1, 2), 32, d(wc), d(bc), 1e-5)).transpose(1, 2).reshape(N, S, Cc)
                scale = max(1.0, ref.abs().max().item())
                e, e3 = (d(y) - ref).abs().max().item() / scale, (d(y3) - ref).abs().max().item() / scale
                assert e <= ulp and e <= 1.5 * e3 + 1e-6, (name, frames, e, e3)
    with pytest.raises(ValueError):
        ops.rows_fused(a, b=b[:, :, :8].contiguous())
    with pytest.raises(ValueError):
        ops.rows_fused(a, base=base)                          # a blend needs its alpha


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 4.0 / 128), (torch.float16, 4.0 / 1024)])
def test_blocks_on_the_token_stream_equal_the_planes_stream(dtype, tol):
    """The token-major residual stream (layers.Tok, MVI_SVD_TOKEN_STREAM): a TimestepEmbedSequential of VideoResBlock (channel-changing
    1x1 skip as a GEMM on rows), SpatialVideoTransformer, VideoResBlock, Downsample, Upsample fed a Tok answers with a Tok whose
    planes equal the same modules fed b c h w (each block's arithmetic is the same; the tails round once either way) and the fp64
    evaluation; the statistics the tails leave are consumed (no standalone statistics for the norms that open blocks 2 and 3); a member
    that knows planes only (a plain ResBlock, a 1x1 zero convolution) is served through planes / as a GEMM on the rows."""
    from multiview_inpaint_amd.svd import hip_ops
    from multiview_inpaint_amd.svd import layers as LY
    from multiview_inpaint_amd.svd.transformer import SpatialVideoTransformer
    g = torch.Generator().manual_seed(29)
    T = 3
    mk_res = lambda cin, cout: LY.VideoResBlock(cin, 256, 0.0, video_kernel_size=[3, 1, 1], out_channels=cout, merge_strategy="learned_with_images",
                                                merge_factor=0.3)
    svt = SpatialVideoTransformer(320, 5, 64, depth=1, context_dim=64, use_linear=True, use_spatial_context=True, time_depth=1, merge_strategy="learned_with_images",
                                  merge_factor=0.5, ff_in=True, attn_mode="softmax-xformers", checkpoint=False)
    seq = LY.TimestepEmbedSequential(mk_res(640, 320), svt, mk_res(320, 320), LY.ResBlock(320, 256, 0.0, out_channels=320),
                                     LY.Downsample(320, True, out_channels=320), LY.Upsample(320, True, out_channels=320),
                                     LY.zero_module(torch.nn.Conv2d(320, 320, 1))).eval()
    with torch.no_grad():
        for p in seq.parameters():
            if p.dim() > 0:
                p.copy_(torch.randn(p.shape, generator=g) * (0.03 if p.dim() > 1 else 0.3))
    x = torch.randn(2 * T, 640, 8, 16, generator=g)
    emb, ctx = torch.randn(2 * T, 256, generator=g), torch.randn(2 * T, 1, 64, generator=g)
    ind = torch.zeros(2, T)
    ind[1, 1] = 1.0
    with torch.no_grad():
        ref = seq.double()(x.double(), emb.double(), ctx.double(), ind.double(), None, T)
        seq = seq.to(dtype).cuda()
        xs, es, cs, inds = x.to(dtype).cuda(), emb.to(dtype).cuda(), ctx.to(dtype).cuda(), ind.cuda()
        old = LY.CONV_N320_MIN_BLOCKS
        try:
            LY.CONV_N320_MIN_BLOCKS = 1
            planes = seq(xs, es, cs, inds, None, T)
            hip_ops.PROFILE = []
            tok = seq(LY.to_tok(xs), es, cs, inds, None, T)
            torch.cuda.synchronize()
            kinds = [rec[0] for rec in hip_ops.PROFILE]
        finally:
            LY.CONV_N320_MIN_BLOCKS = old
            hip_ops.PROFILE = None
    assert isinstance(tok, LY.Tok) and tuple(tok.shape) == tuple(ref.shape) and isinstance(planes, torch.Tensor)
    count = lambda k: sum(1 for x in kinds if x == k)
    # two VideoResBlocks: skip add + blend each; the transformer's exit; nothing of the planes forms of these
    assert count("rows_add") == 3 and count("rows_blend") == 2 and count("tokens_blend_to_planes") == 0 and count("planes_add_to_tokens") == 0, kinds
    # the plain ResBlock went through planes (one pass each way) — and only it
    assert count("planes_to_tokens") == 2 and count("tokens_to_planes_add") >= 1, kinds       # (+ this test's own to_tok)
    got = tok.planes()
    assert rel(got, ref) < tol and rel(planes, ref) < tol and rel(got, planes.double()) < tol, (rel(got, ref), rel(planes, ref), rel(got, planes.double()))


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 4.0 / 128), (torch.float16, 4.0 / 1024)])
def test_video_resblock_on_tokens_equals_the_frames_path(dtype, tol):
    """VideoResBlock (video_model.py:12-81) with the temporal ResBlock evaluated token-major (layers.TIME_STACK_TOKENS: spatial block
    ending on tokens, temporal norms token-major with frame statistics, (3,1,1) convolutions in csrc/linear_n320.hip, blend + b c h w in
    one pass) against the NCHW frames path of the same module and the fp64 evaluation of the reference formulation."""
    from multiview_inpaint_amd.svd import hip_ops
    from multiview_inpaint_amd.svd import layers as LY
    g = torch.Generator().manual_seed(28)
    T = 3
    for cin in (320, 640):
        blk = LY.VideoResBlock(cin, 256, 0.0, video_kernel_size=[3, 1, 1], out_channels=320, merge_strategy="learned_with_images", merge_factor=0.3).eval()
        with torch.no_grad():
            for p in blk.parameters():
                if p.dim() > 0:
                    p.copy_(torch.randn(p.shape, generator=g) * (0.03 if p.dim() > 1 else 0.3))
        x = torch.randn(2 * T, cin, 8, 16, generator=g)
        emb = torch.randn(2 * T, 256, generator=g)
        ind = torch.zeros(2, T)
        ind[1, 1] = 1.0                                                   # one frame treated as an image (alpha = 1 there)
        with torch.no_grad():
            ref = blk.double()(x.double(), emb.double(), T, ind.double())
            blk = blk.to(dtype).cuda()
            xs, es, inds = x.to(dtype).cuda(), emb.to(dtype).cuda(), ind.cuda()
            outs = {}
            old = LY.TIME_STACK_TOKENS, LY.CONV_N320_MIN_BLOCKS
            try:
                LY.CONV_N320_MIN_BLOCKS = 1
                for mode in (False, True):
                    LY.TIME_STACK_TOKENS = mode
                    hip_ops.PROFILE = []
                    outs[mode] = blk(xs, es, T, inds)
                    torch.cuda.synchronize()
                    assert sum(1 for rec in hip_ops.PROFILE if rec[0] == "conv3t_n320") == (2 if mode else 0)
            finally:
                LY.TIME_STACK_TOKENS, LY.CONV_N320_MIN_BLOCKS = old
                hip_ops.PROFILE = None
        assert outs[True].shape == ref.shape and outs[True].is_contiguous()
        assert rel(outs[True], ref) < tol and rel(outs[False], ref) < tol and rel(outs[True], outs[False].double()) < tol


def test_round3_kernels_at_the_full_size_of_the_step():
    """The shapes of the 14-frame 576x1024 step (BASELINE configs[3]), where no fp64 reference fits the time budget: the implicit-GEMM
    convolution against the library's convolution of the same bf16 tensors (level 0 with the 960-channel concatenated input: 495 MB
    of activations, the largest 32-bit lane offsets the kernel sees; level 3 with the K split), linearity of the convolution in its
    input, and the MFMA temporal attention against the fp32-math kernel it replaced."""
    from multiview_inpaint_amd import _lib
    from multiview_inpaint_amd.svd import hip_ops
    g = torch.Generator(device="cuda").manual_seed(77)
    for (N, H, W, C, Co) in [(28, 72, 128, 960, 320), (28, 9, 16, 1280, 1280)]:
        tok = torch.randn(N, H * W, C, device="cuda", generator=g).bfloat16()
        w = (torch.randn(Co, C, 3, 3, device="cuda", generator=g) * (1.0 / (9 * C) ** 0.5)).bfloat16()
        wt = hip_ops.conv3x3_n320_weight(w)
        mine = hip_ops.conv3x3_n320(tok, wt, None, H, W)
        lib = F.conv2d(tok.view(N, H, W, C).permute(0, 3, 1, 2), w.contiguous(memory_format=torch.channels_last), None, 1, 1)
        lib = lib.permute(0, 2, 3, 1).reshape(N, H * W, Co)
        scale = lib.float().abs().max().item()
        assert (mine.float() - lib.float()).abs().max().item() <= scale / 64                        # two bf16 roundings of ~unit values
        assert (mine.float() - lib.float()).pow(2).mean().sqrt().item() <= scale / 1024
        # linearity in the input (exact products, fp32 accumulation): conv(2 x) == 2 conv(x) bit for bit
        assert torch.equal(hip_ops.conv3x3_n320(tok * 2, wt, None, H, W), mine * 2)
        del tok, w, wt, mine, lib
    # temporal attention, level 0: 2 x 9216 x 5 problems of 14 frames
    T, S, Hh = 14, 72 * 128, 5
    qkv = torch.randn(2 * T, S, 3 * Hh * 64, device="cuda", generator=g).bfloat16()
    assert _lib.lib().mvi_attention_temporal_kernel_variant(T, Hh, 64, 1, 3 * Hh * 64, Hh * 64) == 1
    out = hip_ops.attention_temporal_packed(qkv, Hh, T)
    q, k, v = (t.float().contiguous() for t in qkv.chunk(3, dim=-1))                              # fp32 I/O: the fp32-math kernel
    ref = hip_ops.attention_temporal(q, k, v, Hh, T)
    assert (out.float() - ref).abs().max().item() <= 1.0 / 64 and (out.float() - ref).pow(2).mean().sqrt().item() <= 1.0 / 512


def test_round5_kernels_at_the_full_size_of_the_step():
    """The round-5 forms of csrc/linear_n320.hip at the shapes of the 14-frame 576x1024 step (BASELINE configs[3]): the column-group
    projections of levels 1 - 2 and the level-2 GEGLU against the library on the same bf16 tensors (two roundings apart), linearity
    where the arithmetic is exact (x -> 2 x), and the projection with the add + LayerNorm epilogue at the level-0 row count against
    the two kernels it replaces (the residual stream bit-equal, the norm within two ulps of bf16)."""
    from multiview_inpaint_amd.svd import hip_ops
    g = torch.Generator(device="cuda").manual_seed(91)
    for (rows, K, N) in [(28 * 2304, 2560, 640), (28 * 576, 5120, 1280), (28 * 2304, 640, 640)]:
        x = torch.randn(rows, K, device="cuda", generator=g).bfloat16()
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
        b = torch.randn(N, device="cuda", generator=g).bfloat16()
        mine = hip_ops.linear_n320(x, w, b)
        lib = F.linear(x, w, b)
        scale = lib.float().abs().max().item()
        assert (mine.float() - lib.float()).abs().max().item() <= scale / 64
        assert (mine.float() - lib.float()).pow(2).mean().sqrt().item() <= scale / 1024
        assert torch.equal(hip_ops.linear_n320(x * 2, w, None), hip_ops.linear_n320(x, w, None) * 2)
        del x, w, mine, lib
    rows, K, inner = 28 * 576, 1280, 5120                                  # level-2 FeedForward: GEGLU through the n320 form
    x = torch.randn(rows, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(2 * inner, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(2 * inner, device="cuda", generator=g).bfloat16()
    mine = hip_ops.ff_geglu_n320(x, w, b)
    h = F.linear(x, w, b).float()
    ref = h[:, :inner] * F.gelu(h[:, inner:])
    scale = ref.abs().max().item()
    assert (mine.float() - ref).abs().max().item() <= scale / 64 and (mine.float() - ref).pow(2).mean().sqrt().item() <= scale / 1024
    del x, w, h, ref, mine
    rows, K = 28 * 9216, 1280                                              # level-0 FeedForward.net[2] + skip + frame embedding + norm_in
    x = torch.randn(rows, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(320, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(320, device="cuda", generator=g).bfloat16()
    resid = torch.randn(rows, 320, device="cuda", generator=g).bfloat16()
    emb = torch.randn(28, 1, 320, device="cuda", generator=g).bfloat16()
    lw, lb = torch.randn(320, device="cuda", generator=g) * 0.5 + 1.0, torch.randn(320, device="cuda", generator=g) * 0.2
    y, s, s_pre = hip_ops.linear_n320_add_layer_norm(x, w, b, lw, lb, 1e-5, resid=resid, row=emb, ret_pre=True)
    y2, s2, s_pre2 = hip_ops.add_layer_norm(resid, lw, lb, 1e-5, h=hip_ops.linear_n320(x, w, b), row=emb, ret_pre=True)
    assert torch.equal(s, s2) and torch.equal(s_pre, s_pre2)
    assert (y.float() - y2.float()).abs().max().item() <= 2 * 2.0 ** -7 * y2.float().abs().max().item()
