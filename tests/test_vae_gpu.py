"""GPU parity of the first-stage autoencoder on the HIP ops (through the C-ABI): the scaled row
softmax and the wide single-head attention against fp64 PyTorch on the CPU, the small encoder / video decoder against
the golden outputs of the imported reference (tests/golden/vae_small.npz), and the full-size mid-block attention
shape through a size-independent property. fp32 I/O: 1e-4 relative (the reference runs the first stage in fp32:
disable_first_stage_autocast, configs/test/svd_f_est_ctrl_simp1.yaml:6)."""
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
DEV = "cuda:0"


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from multiview_inpaint_amd.svd import hip_ops
    return hip_ops


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "vae_small.npz"))


@pytest.mark.parametrize("rows,cols", [(7, 9216), (33, 77), (5, 13001), (3, 12288), (1, 1), (64, 2304), (2, 12292)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_softmax_rows(ops, rows, cols, dtype):
    g = torch.Generator().manual_seed(rows * 131 + cols)
    x = (torch.randn(rows, cols, generator=g) * 6).to(dtype)
    x[0, cols // 2] = 60.0                                    # one dominant score
    scale = 0.0442
    want = torch.softmax(x.double() * scale, dim=-1)
    got = ops.softmax_rows_(x.to(DEV).clone(), scale)
    tol = {torch.float32: 2e-6, torch.bfloat16: 2 ** -8, torch.float16: 2 ** -11}[dtype]
    assert rel(got, want) < tol * 1.01 + 1e-7
    if dtype == torch.float32:
        assert float((got.double().sum(-1) - 1).abs().max()) < 1e-5


def test_softmax_rows_rejects_bad_arguments(ops):
    x = torch.zeros(4, 8, device=DEV)
    with pytest.raises(Exception):
        ops.softmax_rows_(x, 0.0)
    with pytest.raises(TypeError):
        ops.softmax_rows_(x.t(), 1.0)
    assert ops.softmax_rows_(torch.zeros(0, 8, device=DEV), 1.0).shape == (0, 8)


@pytest.mark.parametrize("B,S,D", [(3, 200, 512), (2, 77, 96), (1, 1024, 128)])
def test_attention_wide_fp32(ops, B, S, D):
    g = torch.Generator().manual_seed(B * S + D)
    q, k, v = (torch.randn(B, S, D, generator=g) for _ in range(3))
    want = F.scaled_dot_product_attention(q.double()[:, None], k.double()[:, None], v.double()[:, None])[:, 0]
    got = ops.attention_wide(q.to(DEV), k.to(DEV), v.to(DEV))
    assert rel(got, want) < 1e-4
    # chunked over the batch (one frame's scores at a time): the same result up to the GEMM library's summation order
    saved = ops._WIDE_SCORE_BYTES
    try:
        ops._WIDE_SCORE_BYTES = S * S * 4
        assert rel(ops.attention_wide(q.to(DEV), k.to(DEV), v.to(DEV)), want) < 1e-4
    finally:
        ops._WIDE_SCORE_BYTES = saved


def test_attention_wide_full_size_row_stochastic(ops):
    """Mid-block shape of the 576x1024 decode (S = 72*128 = 9216, D = 512), one frame: with v = ones the output is 1
    (softmax rows sum to one), and with v = k = one-hot keys the output is the attention matrix's column mass."""
    S, D = 9216, 512
    g = torch.Generator().manual_seed(1)
    q = torch.randn(1, S, D, generator=g).to(DEV)
    k = torch.randn(1, S, D, generator=g).to(DEV)
    out = ops.attention_wide(q, k, torch.ones(1, S, D, device=DEV))
    assert float((out - 1).abs().max()) < 1e-5
    rows = torch.randint(0, S, (64,), generator=g)
    v = torch.randn(1, S, D, generator=g).to(DEV)
    got = ops.attention_wide(q, k, v)[0, rows.to(DEV)]
    want = F.scaled_dot_product_attention(q[0, rows.to(DEV)].double()[None, None], k.double()[:, None], v.double()[:, None])[0, 0]
    assert rel(got, want) < 1e-4


@pytest.fixture(scope="module")
def engine():
    from sgm.util import instantiate_from_config
    import test_vae_cpu as C
    eng = instantiate_from_config(C.FIRST_STAGE).eval()
    eng.encoder.load_state_dict(H.seeded_state_dict(eng.encoder, 41), strict=True)
    eng.decoder.load_state_dict(H.seeded_state_dict(eng.decoder, 42), strict=True)
    return eng.to(DEV)


def test_small_autoencoder_on_hip_ops_matches_the_reference_golden(G, engine, ops):
    ops.PROFILE = []
    try:
        with torch.no_grad():
            m, _ = engine.encode(H.vae_inputs(31).to(DEV), unregularized=True)
            z = torch.tensor(G["z_sample"]).to(DEV)
            y = engine.decode(z, timesteps=H.VAE_T)
            y2 = engine.decode(torch.cat([z, z.flip(0)]), timesteps=H.VAE_T)
            ys = engine.decode(z, timesteps=H.VAE_T, skip_video=True)
        torch.cuda.synchronize()
        kinds = set(k for k, *_ in ops.PROFILE)
    finally:
        ops.PROFILE = None
    assert rel(m, G["enc_moments"]) < 1e-4
    assert rel(y, G["vdec_out_conv_only"]) < 1e-4
    assert rel(y2, G["vdec_out2_conv_only"]) < 1e-4
    assert rel(ys, G["vdec_out_skip_video"]) < 1e-4
    # the HIP kernels are what ran
    assert {"groupnorm", "bias_residual", "softmax_rows", "tokens_to_planes_add"} <= kinds, kinds


def test_video_block_on_hip_ops(G):
    from sgm.modules.autoencoding.temporal_ae import VideoBlock
    vb = VideoBlock(64).eval()
    vb.load_state_dict(H.seeded_state_dict(vb, 44), strict=True)
    vb = vb.to(DEV)
    xb = torch.randn(2 * H.VAE_T, 64, 8, 4, generator=torch.Generator().manual_seed(32)).to(DEV)
    with torch.no_grad():
        assert rel(vb(xb, timesteps=H.VAE_T), G["vblock_out"]) < 1e-4
        assert rel(vb(xb, timesteps=H.VAE_T, skip_video=True), G["vblock_out_skip"]) < 1e-4


def test_full_size_first_stage_matches_the_reference(golden_dir, ops):
    """The first stage at its FULL size (configs/test/svd_f_est_ctrl_simp1.yaml:131-159: ch 128, ch_mult [1, 2, 4, 4], 2 ResBlocks per
    level) on 576x1024 frames against the imported reference's fp32 CPU outputs (tests/golden/vae_full.npz,
    tools/gen_golden_vae_full.py): the video decoder on two 72x128 latent frames — final frames (subsampled + one dense crop), the
    mid-block attention output (S = 9216, D = 512: attention_wide's production shape) and the last block of the 288x512 level —
    and the encoder's moments for one frame. fp32 I/O: 1e-4 relative, the module's bar."""
    from sgm.util import instantiate_from_config
    G = np.load(os.path.join(golden_dir, "vae_full.npz"))
    cfg = {"target": "sgm.models.autoencoder.AutoencodingEngine", "params": {
        "loss_config": {"target": "torch.nn.Identity"},
        "regularizer_config": {"target": "sgm.modules.autoencoding.regularizers.DiagonalGaussianRegularizer"},
        "encoder_config": {"target": "sgm.modules.diffusionmodules.model.Encoder", "params": H.FULL_VAE},
        "decoder_config": {"target": "sgm.modules.autoencoding.temporal_ae.VideoDecoder",
                           "params": dict(H.FULL_VAE, video_kernel_size=[3, 1, 1])}}}
    eng = instantiate_from_config(cfg).eval()
    assert sorted(eng.encoder.state_dict().keys()) == list(G["enc_keys"])
    assert sorted(eng.decoder.state_dict().keys()) == list(G["vdec_keys"])
    eng.encoder.load_state_dict(H.seeded_state_dict(eng.encoder, 51), strict=True)
    eng.decoder.load_state_dict(H.seeded_state_dict(eng.decoder, 52), strict=True)
    eng = eng.to(DEV)
    acts = {}
    hooks = [eng.decoder.mid.attn_1.register_forward_hook(lambda m, i, o: acts.__setitem__("mid_attn", o)),
             eng.decoder.up[1].block[2].register_forward_hook(lambda m, i, o: acts.__setitem__("up1", o))]
    ops.PROFILE = []
    try:
        with torch.no_grad():
            y = eng.decode(H.vae_full_latent(61).to(DEV), timesteps=H.FULL_VAE_T)
            m, _ = eng.encode(H.vae_inputs(62, T=1, hw=H.FULL_VAE_HW).to(DEV), unregularized=True)
        torch.cuda.synchronize()
        kinds = set(k for k, *_ in ops.PROFILE)
    finally:
        ops.PROFILE = None
        for h in hooks:
            h.remove()
    assert tuple(y.shape) == (H.FULL_VAE_T, 3) + H.FULL_VAE_HW

    def rel_to(a, want, absmax):
        return float((a.detach().double().cpu() - torch.as_tensor(want).double()).abs().max() / absmax)
    errs = {"mid_attn": rel_to(acts["mid_attn"][:, ::8, ::4, ::4], G["vdec_mid_attn_sub"], float(G["vdec_mid_attn_absmax"])),
            "up1": rel_to(acts["up1"][:, ::16, ::8, ::8], G["vdec_up1_sub"], float(G["vdec_up1_absmax"])),
            "out_sub": rel_to(y[H.FULL_VAE_SUB], G["vdec_out_sub"], float(G["vdec_out_absmax"])),
            "out_crop": rel_to(y[H.FULL_VAE_CROP], G["vdec_out_crop"], float(G["vdec_out_absmax"])),
            "enc_moments": rel(m, G["enc_moments"])}
    print("full-size first stage, relative max errors:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert abs(float(y.double().mean()) - float(G["vdec_out_mean"])) < 1e-4 * float(G["vdec_out_absmax"])
    assert max(errs.values()) < 1e-4, errs
    assert {"groupnorm", "bias_residual", "softmax_rows", "tokens_to_planes_add"} <= kinds, kinds
    # round 6: the DEFAULT decode (the fp32 contract above) ran its convolutions with split operands on the bf16 matrix pipe
    assert {"conv_split3", "groupnorm_split"} <= kinds, kinds
    H.report(f"full-size first stage (decoder convolutions on the bf16 matrix pipe, split operands): relative max errors { {k: f'{v:.2e}' for k, v in errs.items()} } (bar 1e-4)")


@pytest.mark.parametrize("dt,bar", [(torch.bfloat16, 1.0), (torch.float16, 0.25)])
def test_full_size_first_stage_reduced_precision_decode_within_the_reference_autocast_budget(golden_dir, ops, dt, bar):
    """decode_first_stage(dtype=...) — the opt-in reduced-precision decode — at the full size of configs[3], two 72x128 latent frames ->
    576x1024 frames, against the imported reference's fp32 frames (tests/golden/vae_full.npz). Budget: the error of the reference's OWN
    bf16-autocast decode of the same weights and latents against its fp32 decode (tools/gen_golden_vae_full_bf16.py; max 1.07e-2, rms
    1.8e-3 of the largest output). Since round 6 the reduced-precision decode is the token-major walk of svd/vae_split.py with ONE
    rounded value per convolution operand (rounded products, fp32 accumulation: an autocast convolution's arithmetic) and the residual
    stream, norms and attention in fp32 — what the reference's autocast keeps in fp32 too. bf16: within 1.0 x of the reference's own
    autocast error (observed 0.55 x; the bf16 COPY of the decoder of round 5 measured 1.55 - 1.87 x: its residual stream was bf16);
    f16 (11-bit mantissa): within 0.25 x (observed 0.08 x; round 5: 0.24 x). The fp32 default is a different contract
    (test_full_size_first_stage_matches_the_reference)."""
    import svd_helpers as H
    from sgm.util import instantiate_from_config
    from multiview_inpaint_amd.svd import vae
    G = np.load(os.path.join(golden_dir, "vae_full.npz"))
    assert "budget_bf16_out_sub" in G.files, "run tools/gen_golden_vae_full_bf16.py (build container only)"
    cfg = {"target": "sgm.models.autoencoder.AutoencodingEngine", "params": {
        "loss_config": {"target": "torch.nn.Identity"},
        "regularizer_config": {"target": "sgm.modules.autoencoding.regularizers.DiagonalGaussianRegularizer"},
        "encoder_config": {"target": "sgm.modules.diffusionmodules.model.Encoder", "params": H.FULL_VAE},
        "decoder_config": {"target": "sgm.modules.autoencoding.temporal_ae.VideoDecoder",
                           "params": dict(H.FULL_VAE, video_kernel_size=[3, 1, 1])}}}
    eng = instantiate_from_config(cfg).eval()
    eng.decoder.load_state_dict(H.seeded_state_dict(eng.decoder, 52), strict=True)
    eng = eng.to(DEV)
    z = H.vae_full_latent(61).to(DEV)
    y = vae.decode_first_stage(eng, z * 0.18215, dtype=dt)              # (decode_first_stage unscales by 1 / 0.18215)
    torch.cuda.synchronize()
    assert y.dtype == torch.float32 and tuple(y.shape) == (H.FULL_VAE_T, 3) + H.FULL_VAE_HW
    assert next(eng.decoder.parameters()).dtype == torch.float32        # the model itself stays fp32
    amax = float(G["vdec_out_absmax"])
    worst = 0.0
    for name, sl in (("out_sub", H.FULL_VAE_SUB), ("out_crop", H.FULL_VAE_CROP)):
        d = (y[sl].double().cpu() - torch.as_tensor(G["vdec_" + name]).double()).abs()
        e_max, e_rms = float(d.max()) / amax, float(d.pow(2).mean().sqrt()) / amax
        b_max, b_rms = (float(v) for v in G["budget_bf16_" + name])
        worst = max(worst, e_max / b_max, e_rms / b_rms)
        assert e_max <= bar * b_max and e_rms <= bar * b_rms, (name, e_max, b_max, e_rms, b_rms)
    H.report(f"first-stage decode at full size in {dt}: worst ratio of its error to the reference's own bf16-autocast decode error = {worst:.2f}")


@pytest.mark.parametrize("N,H,W,C,Co,taps", [(2, 24, 32, 128, 128, 9), (3, 9, 15, 64, 320, 9), (1, 40, 33, 256, 512, 9), (2, 16, 16, 512, 256, 9),
                                             (2, 7, 300, 128, 128, 3), (1, 14, 256, 256, 256, 3), (3, 2, 40, 64, 64, 3)])
def test_split_operand_convolution_has_fp32_accuracy(ops, N, H, W, C, Co, taps):
    """mvi_conv3x3_split3_f32 / mvi_conv3t_split3_f32 behind hip_ops.conv_split3 (round 6): x . w ~= x_hi . w_hi + x_hi . w_lo + x_lo . w_hi
    on the bf16 matrix pipe, fp32 accumulate, against F.conv2d / F.conv3d of the SAME fp32 operands in fp64 — 3x3 / padding 1 over
    images, (3,1,1) / padding (1,0,0) over frames; C_out = 128 (padded to a 256-column group), 256, 512 (two groups), 320 (the 320-column
    form), 64; ragged row counts; and the batch cut into launches (forced small here). Bar 3e-5 of the output scale: the dropped
    x_lo . w_lo terms are 2^-16 of a product each (observed ~1e-5) — an order under the 1e-4 the decoder is held to."""
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + C + taps)
    x = torch.randn(N, H * W, C, generator=g) * torch.rand(1, 1, C, generator=g).mul(3).add(0.2)      # channels of different scales
    if taps == 9:
        w = torch.randn(Co, C, 3, 3, generator=g) * (9 * C) ** -0.5
        want = F.conv2d(x.double().view(N, H, W, C).permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1).reshape(N, H * W, Co)
    else:                                                           # N videos of H frames of W pixels
        w = torch.randn(Co, C, 3, 1, 1, generator=g) * (3 * C) ** -0.5
        x5 = x.double().view(N, H, W, C).permute(0, 3, 1, 2)[..., None]                                  # b c t (pixels) 1
        want = F.conv3d(x5, w.double(), padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(N, H * W, Co)
    xd = x.to(DEV)
    x2 = ops.group_norm_split(xd.view(N, H * W, C) if taps == 9 else xd.view(N * H, W, C), 0, None, None, 0.0, False)
    # the one-value forms (the opt-in reduced-precision decode): products of ROUNDED operands, accumulated in fp32 — against fp64 on the same
    # rounded operands to fp32 summation accuracy
    for mode, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        x1 = ops.group_norm_split(xd.view(N, H * W, C), 0, None, None, 0.0, False, mode=mode)
        assert x1.dtype == dt and torch.equal(x1, xd.to(dt))
        g1 = ops.conv_split3(x1.reshape(-1, C), ops.split3_weight(w.to(DEV), mode), N, H, W, Co, taps=taps, mode=mode).view(N, H * W, Co)
        xr, wr = x.to(dt).double(), w.to(dt).double()
        if taps == 9:
            w1 = F.conv2d(xr.view(N, H, W, C).permute(0, 3, 1, 2), wr, padding=1).permute(0, 2, 3, 1).reshape(N, H * W, Co)
        else:
            w1 = F.conv3d(xr.view(N, H, W, C).permute(0, 3, 1, 2)[..., None], wr, padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(N, H * W, Co)
        assert rel(g1, w1) < 5e-6, (mode, rel(g1, w1))
    hi, lo = ops.split_hi_lo(xd)
    assert torch.equal(x2[..., :C].reshape(N, H * W, C), hi) and torch.equal(x2[..., C:].reshape(N, H * W, C), lo)
    w3 = ops.split3_weight(w.to(DEV))
    assert w3.shape == (-(-Co // ops._lib.lib().mvi_conv_split3_group(Co)) * ops._lib.lib().mvi_conv_split3_group(Co), taps * 3 * C)
    got = ops.conv_split3(x2.reshape(-1, 2 * C), w3, N, H, W, Co, taps=taps).view(N, H * W, Co)
    torch.cuda.synchronize()
    e = rel(got, want)
    assert e < 3e-5, e
    if N > 1:                                                       # one image / video per launch: the same values
        cut = ops.conv_split3(x2.reshape(-1, 2 * C), w3, N, H, W, Co, taps=taps, _max_bytes=H * W * 4 * C).view(N, H * W, Co)
        assert torch.equal(cut, got)
    if taps == 3:                                                   # a video larger than one launch may address: the pixel axis is cut
        cut = ops.conv_split3(x2.reshape(-1, 2 * C), w3, N, H, W, Co, taps=3, _max_bytes=H * W * 4 * C // 3 + 1).view(N, H * W, Co)
        assert torch.equal(cut, got)
    assert rel(got, want) < 0.02 * rel(F.conv2d(hi.float().view(N, H, W, C).permute(0, 3, 1, 2), w.to(DEV).bfloat16().float(), padding=1)
                                       .permute(0, 2, 3, 1).reshape(N, H * W, Co), want) if taps == 9 else True   # (two orders better than plain bf16)


@pytest.mark.parametrize("N,S,C,frames,silu", [(4, 300, 128, 1, True), (6, 64, 256, 3, True), (2, 1000, 512, 2, False), (3, 77, 64, 1, True)])
def test_group_norm_with_split_output(ops, N, S, C, frames, silu):
    """mvi_groupnorm_silu_tok2tok_split: GroupNorm(32, eps 1e-6)(+SiLU) of an fp32 token-major tensor with per-sample channel bias,
    statistics per video of `frames` samples, written as (hi | lo) bf16 halves: hi + lo against fp64 to 2^-15 of the scale (two bf16
    roundings), |lo| <= half an ulp of hi."""
    g = torch.Generator().manual_seed(N + S + C)
    x = torch.randn(N, S, C, generator=g) * 2 + 0.5
    wgt, b = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    cb = 0.3 * torch.randn(N, C, generator=g)
    xf = (x.double() + cb.double()[:, None, :]).reshape(N // frames, frames * S, C).transpose(1, 2)
    want = F.group_norm(xf, 32, wgt.double(), b.double(), 1e-6)
    want = (F.silu(want) if silu else want).transpose(1, 2).reshape(N, S, C)
    y2 = ops.group_norm_split(x.to(DEV), 32, wgt.to(DEV), b.to(DEV), 1e-6, silu, chan_bias=cb.to(DEV), frames=frames)
    torch.cuda.synchronize()
    hi, lo = y2[..., :C].float(), y2[..., C:].float()
    assert rel(hi + lo, want) < 2.0 ** -15
    assert float((lo.abs() / hi.abs().clamp_min(1e-30)).max()) <= 2.0 ** -8         # lo is the rounding residue of hi: at most half an ulp of it


def test_rows_axpb(ops):
    """a + alpha (b + bias[c]) on fp32 token rows in one pass, also in place into either operand (bit-exact against the same fp32 formula)."""
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(3, 77, 128, generator=g).to(DEV), torch.randn(3, 77, 128, generator=g).to(DEV)
    bias = torch.randn(128, generator=g).to(DEV)
    want = a + 0.37 * (b + bias)
    got = ops.rows_axpb(a, b.clone(), bias, alpha=0.37)
    assert rel(got, want) < 1e-6
    b2 = b.clone()
    assert ops.rows_axpb(a, b2, bias, alpha=0.37) is b2 and torch.equal(b2, got)
    a2 = a.clone()
    assert torch.equal(ops.rows_axpb(a2, b, None, alpha=1.0, out=a2), a + b)
