"""CPU parity of the denoise-loop restatement (multiview_inpaint_amd.svd, imported through the
drop-in `sgm` / `models.csvd` names) against golden outputs of the REFERENCE modules
(tests/golden/sgm_small.npz, made by tools/gen_golden_sgm.py from /root/reference).
Tolerance 1e-4 relative (BASELINE.json north_star); fp32 CPU results agree to ~1e-6."""
import os
import sys

import numpy as np
import pytest
import torch

import svd_helpers as H

DROPIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multiview_inpaint_amd", "dropin")
if DROPIN not in sys.path:
    sys.path.insert(0, DROPIN)

from sgm.util import instantiate_from_config  # noqa: E402

RTOL = 1e-4


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "sgm_small.npz"))


@pytest.fixture(scope="module")
def nets():
    unet = instantiate_from_config({"target": "sgm.modules.diffusionmodules.video_model.VideoUNet", "params": H.SMALL_UNET}).eval()
    cunet = instantiate_from_config({"target": "models.csvd.ControlledVideoUNet", "params": H.SMALL_UNET}).eval()
    cnet = instantiate_from_config({"target": "models.csvd.ControlNet", "params": H.SMALL_CTRL}).eval()
    unet.load_state_dict(H.seeded_state_dict(unet, 11), strict=True)
    cunet.load_state_dict(H.seeded_state_dict(cunet, 11), strict=True)
    cnet.load_state_dict(H.seeded_state_dict(cnet, 12), strict=True)
    return unet, cunet, cnet


def test_scalar_known_answers(G):
    from sgm.modules.diffusionmodules.discretizer import EDMDiscretization
    from sgm.modules.diffusionmodules.denoiser_scaling import VScalingWithEDMcNoise
    from sgm.modules.diffusionmodules.guiders import LinearPredictionGuider
    from sgm.modules.diffusionmodules.util import timestep_embedding
    sig = EDMDiscretization(sigma_min=0.002, sigma_max=700.0, rho=7.0)(25)
    np.testing.assert_allclose(sig.numpy(), G["sigmas25"], rtol=1e-6)
    # SURVEY.md §8c captured values
    assert sig.shape == (26,) and sig[-1] == 0
    np.testing.assert_allclose(sig[[0, 1, 2, 12, 23, 24]].numpy(),
                               [700.000122, 545.729492, 421.569122, 15.589973, 0.007883, 0.002], rtol=2e-4)
    s = torch.tensor(G["scaling_sigma"])
    np.testing.assert_allclose(torch.stack(VScalingWithEDMcNoise()(s)).numpy(), G["scaling_out"], rtol=1e-6)
    np.testing.assert_allclose([float(v[0]) for v in VScalingWithEDMcNoise()(torch.tensor([700.0]))],
                               [2.0408122e-06, -0.99999893, 1.4285699e-03, 1.63777006], rtol=1e-5)
    g = LinearPredictionGuider(max_scale=2.5, num_frames=14, min_scale=1.0)
    np.testing.assert_array_equal(g.scale.numpy(), G["guider_scale14"])
    e = timestep_embedding(torch.tensor([0.25 * np.log(700.0), 0.0, -1.3]).float(), 320)
    np.testing.assert_allclose(e.numpy(), G["temb_320"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(e[0, :3].numpy(), [-0.0669237, 0.0246392, 0.1109036], atol=2e-6)   # cos half first
    np.testing.assert_allclose(timestep_embedding(torch.tensor([3.0, 0.5]), 33, max_period=100).numpy(), G["temb_odd"],
                               rtol=1e-6, atol=1e-7)


def test_state_dict_keys_match_reference(G, nets):
    unet, cunet, cnet = nets
    assert sorted(unet.state_dict().keys()) == list(G["unet_keys"])
    assert sorted(cunet.state_dict().keys()) == list(G["unet_keys"])
    assert sorted(cnet.state_dict().keys()) == list(G["cnet_keys"])


def test_videounet_and_activations(G, nets):
    unet, _, _ = nets
    inp = H.seeded_inputs(21)
    kw = dict(num_video_frames=H.T_FRAMES, image_only_indicator=inp["image_only_indicator"])
    xin = torch.cat([inp["x"], inp["concat"]], 1)
    tt = 0.25 * inp["sigma"].log()
    acts = {}
    hooks = [unet.input_blocks[1].register_forward_hook(lambda m, i, o: acts.__setitem__("in1", o)),
             unet.input_blocks[3].register_forward_hook(lambda m, i, o: acts.__setitem__("in3", o)),
             unet.middle_block.register_forward_hook(lambda m, i, o: acts.__setitem__("mid", o)),
             unet.output_blocks[0].register_forward_hook(lambda m, i, o: acts.__setitem__("out0", o))]
    with torch.no_grad():
        y = unet(xin, tt, inp["crossattn"], inp["vector"], **kw)
        for h in hooks:
            h.remove()
        y1 = unet(xin, tt, inp["crossattn"], inp["vector"], num_video_frames=H.T_FRAMES,
                  image_only_indicator=torch.ones(1, H.T_FRAMES))
    for k, v in acts.items():
        assert rel(v.numpy(), G["unet_act_" + k]) < RTOL, k
    assert rel(y.numpy(), G["unet_out"]) < RTOL
    assert rel(y1.numpy(), G["unet_out_imageonly"]) < RTOL


def test_controlnet_and_controlled_unet(G, nets):
    _, cunet, cnet = nets
    inp = H.seeded_inputs(21)
    kw = dict(num_video_frames=H.T_FRAMES, image_only_indicator=inp["image_only_indicator"])
    xin = torch.cat([inp["x"], inp["concat"]], 1)
    tt = 0.25 * inp["sigma"].log()
    with torch.no_grad():
        ctrls = cnet(xin, inp["control_hint"], tt, inp["crossattn"], inp["vector"], **kw)
        n = len(ctrls)
        for i, c in enumerate(ctrls):
            assert rel(c.numpy(), G[f"ctrl_{i}"]) < RTOL, i
        lst = [c.clone() for c in ctrls]
        y = cunet(xin, tt, inp["crossattn"], inp["vector"], control=lst, **kw)
    assert len(lst) == 0 and n == 5                       # the caller's list is consumed (csvd.py:80,91)
    assert rel(y.numpy(), G["cunet_out"]) < RTOL


def test_denoiser_forward(G, nets):
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    unet, _, _ = nets
    inp = H.seeded_inputs(21)
    den = Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    cond = dict(crossattn=inp["crossattn"], vector=inp["vector"], concat=inp["concat"])
    with torch.no_grad():
        d = den(OpenAIWrapper(unet), inp["x"], inp["sigma"], cond, num_video_frames=H.T_FRAMES,
                image_only_indicator=inp["image_only_indicator"])
    assert rel(d.numpy(), G["denoiser_out"]) < RTOL


def test_euler_sampler_trajectory_with_controlnet(G, nets):
    """5 Euler steps, per-frame linear guidance, control_hint doubled with the batch — through the
    Lightning-free engine that stands in for SVDEngine.apply_model/sample (csvd.py:1086-1152)."""
    from models.csvd import SVDInpaintEngine
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    _, cunet, cnet = nets
    T = H.T_FRAMES
    one = H.seeded_inputs(22, cfg_doubled=False)
    sampler = instantiate_from_config({
        "target": "sgm.modules.diffusionmodules.sampling.EulerEDMSampler",
        "params": {"num_steps": 5, "device": "cpu",
                   "discretization_config": {"target": "sgm.modules.diffusionmodules.discretizer.EDMDiscretization",
                                             "params": {"sigma_max": 700.0}},
                   "guider_config": {"target": "sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                                     "params": {"max_scale": 2.5, "min_scale": 1.0, "num_frames": T,
                                                "additional_cond_keys": ["control_hint"]}}}})
    eng = SVDInpaintEngine(cunet, cnet, Denoiser({"target": "sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}),
                           sampler, control_scales=[1.0] * 5)
    c = dict(crossattn=one["crossattn"], vector=one["vector"], concat=one["concat"], control_hint=one["control_hint"])
    uc = dict(crossattn=torch.zeros_like(one["crossattn"]), vector=torch.zeros_like(one["vector"]),
              concat=torch.zeros_like(one["concat"]), control_hint=one["control_hint"])
    traj = []

    def denoiser(x, sigma, cc):
        d = eng.denoise(x, sigma, cc, num_video_frames=T, image_only_indicator=one["image_only_indicator"])
        traj.append(d.clone())
        return d
    with torch.no_grad():
        xs = sampler(denoiser, one["x"].clone(), c, uc=uc)
    assert len(traj) == 5 and traj[0].shape[0] == 2 * T
    assert rel(traj[0].numpy(), G["sample_denoised_step0"]) < RTOL
    assert rel(traj[4].numpy(), G["sample_denoised_step4"]) < 5 * RTOL
    assert rel(xs.numpy(), G["sample_final"]) < 5 * RTOL
    # the same loop with the hint stem cached per sample (what SVDInpaintEngine.sample() does): identical trajectory,
    # the stem evaluated once instead of once per step
    calls, stem = [0], cnet._hint_stem

    def counted(*a, **k):
        calls[0] += 1
        return stem(*a, **k)
    cnet._hint_stem = counted
    try:
        first, traj[:] = list(traj), []
        with torch.no_grad(), cnet.hint_cache():
            xs2 = sampler(denoiser, one["x"].clone(), c, uc=uc)
    finally:
        del cnet._hint_stem
    assert calls[0] == 1 and torch.equal(xs2, xs) and all(torch.equal(a, b) for a, b in zip(first, traj))
    assert "_hint_slot" not in cnet.__dict__
    # SVDInpaintEngine.sample(): draws the start from the global RNG (csvd.py:1269), runs the same loop inside the cache
    torch.manual_seed(5)
    xs3 = eng.sample(None, c, uc=uc, batch_size=one["x"].shape[0], shape=tuple(one["x"].shape[1:]), num_video_frames=T,
                     image_only_indicator=one["image_only_indicator"])
    assert "_hint_slot" not in cnet.__dict__ and "_cond_cache" not in sampler.guider.__dict__      # released with the sample
    torch.manual_seed(5)
    with torch.no_grad():
        xs4 = sampler(denoiser, torch.randn(*one["x"].shape), c, uc=uc)
    assert torch.equal(xs3, xs4)


def test_full_size_parameter_counts_on_meta():
    """The SVD configuration of the reference YAMLs: 1 524 623 082 UNet and
    682 022 113 ControlNet parameters — counted on the meta device, nothing is allocated."""
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from models.csvd import ControlNet
    cfg = dict(in_channels=8, out_channels=4, model_channels=320, channel_mult=[1, 2, 4, 4], num_res_blocks=2,
               attention_resolutions=[4, 2, 1], num_head_channels=64, transformer_depth=1, context_dim=1024,
               adm_in_channels=768, num_classes="sequential", use_linear_in_transformer=True, extra_ff_mix_layer=True,
               use_spatial_context=True, merge_strategy="learned_with_images", video_kernel_size=[3, 1, 1],
               use_checkpoint=True, spatial_transformer_attn_type="softmax-xformers")
    with torch.device("meta"):
        unet = VideoUNet(**cfg)
        ccfg = {k: v for k, v in cfg.items() if k != "out_channels"}
        cnet = ControlNet(hint_channels=7, **ccfg)
    assert sum(p.numel() for p in unet.parameters()) == 1_524_623_082
    assert sum(p.numel() for p in cnet.parameters()) == 682_022_113
    assert len(cnet.zero_convs) == 12
    n_attn = sum(1 for m in unet.modules() if type(m).__name__ == "SpatialVideoTransformer")
    n_res = sum(1 for m in unet.modules() if type(m).__name__ == "VideoResBlock")
    assert (n_attn, n_res) == (16, 22)


def test_single_key_cross_attention_shortcut_is_exact():
    """S_k = 1: the value-row shortcut equals softmax attention bit for bit up to the out-projection."""
    from sgm.modules.attention import CrossAttention
    torch.manual_seed(0)
    att = CrossAttention(query_dim=32, context_dim=24, heads=4, dim_head=8).eval()
    x, ctx = torch.randn(3, 10, 32), torch.randn(3, 1, 24)
    with torch.no_grad():
        fast = att(x, context=ctx)
        q, k, v = att.to_q(x), att.to_k(ctx), att.to_v(ctx)
        qh, kh, vh = (t.reshape(3, -1, 4, 8).transpose(1, 2) for t in (q, k, v))
        ref = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(3, 10, 32)
        ref = att.to_out(ref)
    assert torch.allclose(fast, ref, atol=1e-6)
    with pytest.raises(NotImplementedError):
        att(x, context=ctx, mask=torch.ones(3, 10, 1, dtype=torch.bool))


def test_residual_and_layernorm_riding_on_the_projections_equal_the_reference_block_order():
    """Round 5: every inner residual add + the LayerNorm behind it is asked for through the output projection of the layer that
    produced the addend (CrossAttention / FeedForward `fuse=`, one kernel on the GPU at the level-0 width; here the two-step fallback),
    the stem's proj_in carries norm1 and the spatial FeedForward carries the temporal block's entry. The composed result equals the
    reference's own order of operations (attention.py:544-572 `x = attn1(norm1(x)) + x; x = attn2(norm2(x)) + x; ff(norm3(x)) + x`;
    video_attention.py:278-296 spatial block, + frame embedding, temporal block, blend) on the CPU: one and several context tokens."""
    from multiview_inpaint_amd.svd import transformer as T
    torch.manual_seed(0)
    blk = T.BasicTransformerBlock(64, 2, 32, context_dim=48).eval()
    x = torch.randn(4, 10, 64)
    with torch.no_grad():
        for ctx in (torch.randn(4, 1, 48), torch.randn(4, 3, 48)):
            h, skip = blk.forward_deferred(x, ctx)
            ref = blk.attn1(blk.norm1(x)) + x
            ref = blk.attn2(blk.norm2(ref), context=ctx) + ref
            ref = blk.ff(blk.norm3(ref)) + ref
            assert torch.allclose(h + skip, ref, atol=1e-6)
            # the caller's next add + norm on the FeedForward, the caller's norm1 handed in
            nxt = torch.nn.LayerNorm(64)
            emb = torch.randn(4, 1, 64)
            y, s, s_pre = blk.forward_deferred(x, ctx, n1=blk.norm1(x), ff_fuse=dict(norm=nxt, row=emb, ret_pre=True))
            assert torch.allclose(s_pre, ref, atol=1e-6) and torch.allclose(s, ref + emb, atol=1e-6) and torch.allclose(y, nxt(ref + emb), atol=1e-5)
        vt = T.SpatialVideoTransformer(64, 2, 32, depth=1, use_linear=True, context_dim=48, use_spatial_context=True, timesteps=2,
                                       merge_strategy="learned_with_images", ff_in=True, time_depth=1).eval()
        for p_ in vt.parameters():
            torch.nn.init.normal_(p_, std=0.1)
        xv, ind, ctx = torch.randn(4, 64, 4, 4), torch.zeros(2, 2), torch.randn(4, 1, 48)
        got = vt(xv, context=ctx, timesteps=2, image_only_indicator=ind)          # the in-place route with every fuse request
        t = vt._tokens_in(xv)
        emb = vt.time_pos_embed(vt._frame_embedding(2, 2, xv.device))[:, None, :]
        t1 = vt.transformer_blocks[0](t, context=ctx)
        tt = vt.time_stack[0](t1 + emb, context=ctx[::2].repeat_interleave(16, dim=0), timesteps=2)
        ref = vt._tokens_out(vt.time_mixer(x_spatial=t1, x_temporal=tt, image_only_indicator=ind), xv)
        assert torch.allclose(got, ref, atol=2e-6)


def test_step_invariant_caches_follow_their_inputs():
    """The per-step tensors that depend only on shapes or on rarely-changing inputs are cached (blend factors of
    AlphaBlender, the sinusoidal frequency table, the frame-index embedding): the cached values equal the uncached
    formulas (util.py:207-231, :343-356) and follow in-place changes of their inputs."""
    import math

    from multiview_inpaint_amd.svd.layers import AlphaBlender, timestep_embedding
    t = torch.tensor([0.0, 3.5, 999.0])
    for _ in range(2):                                        # second call is served from the table cache
        got = timestep_embedding(t, 320)
        freqs = torch.exp(-math.log(10000) * torch.arange(160, dtype=torch.float32) / 160)
        ang = t[:, None].float() * freqs[None]
        assert torch.equal(got, torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1))
    ab = AlphaBlender(0.3, "learned_with_images", "b t -> b 1 t 1 1")
    ind = torch.zeros(2, 3)
    ind[1, 2] = 1.0

    def formula(i):
        return torch.where(i.bool(), torch.ones(1, 1), torch.sigmoid(ab.mix_factor.detach())[..., None]).reshape(2, 1, 3, 1, 1)
    with torch.no_grad():
        a0 = ab.get_alpha(ind)
        assert torch.equal(a0, formula(ind)) and ab.get_alpha(ind) is a0          # hit
        ind[0, 0] = 1.0                                                            # in-place change of the indicator
        a1 = ab.get_alpha(ind)
        assert a1 is not a0 and torch.equal(a1, formula(ind))
        ab.mix_factor.add_(0.5)                                                    # an optimizer step
        a2 = ab.get_alpha(ind)
        assert a2 is not a1 and torch.equal(a2, formula(ind))
        other = ind.clone()
        other[1, 1] = 1.0
        assert torch.equal(ab.get_alpha(other), formula(other))                    # another tensor
    a3 = ab.get_alpha(ind)                                                         # autograd on: never cached, differentiable
    assert a3.requires_grad and torch.equal(a3.detach(), formula(ind))


def test_head_dim_64_config_matches_reference_golden(golden_dir):
    """The production head width (num_head_channels 64) on a 16x16 latent: CPU fp32 restatement vs the reference's fp32
    outputs (tests/golden/sgm_hd64.npz, tools/gen_golden_sgm_hd64.py). The GPU suite runs the same nets in bf16 through
    the MFMA attention kernel and holds them to the reference's own autocast error recorded in the same fixture."""
    G64 = np.load(os.path.join(golden_dir, "sgm_hd64.npz"))
    unet = instantiate_from_config({"target": "sgm.modules.diffusionmodules.video_model.VideoUNet", "params": H.SMALL_UNET64}).eval()
    cunet = instantiate_from_config({"target": "models.csvd.ControlledVideoUNet", "params": H.SMALL_UNET64}).eval()
    cnet = instantiate_from_config({"target": "models.csvd.ControlNet", "params": H.SMALL_CTRL64}).eval()
    unet.load_state_dict(H.seeded_state_dict(unet, 31), strict=True)
    cunet.load_state_dict(H.seeded_state_dict(cunet, 31), strict=True)
    cnet.load_state_dict(H.seeded_state_dict(cnet, 32), strict=True)
    inp = H.seeded_inputs(41, hw=H.LATENT_HW64, cfg=H.SMALL_UNET64)
    kw = dict(num_video_frames=H.T_FRAMES, image_only_indicator=inp["image_only_indicator"])
    xin = torch.cat([inp["x"], inp["concat"]], 1)
    tt = 0.25 * inp["sigma"].log()
    with torch.no_grad():
        y = unet(xin, tt, inp["crossattn"], inp["vector"], **kw)
        ctrls = cnet(xin, inp["control_hint"], tt, inp["crossattn"], inp["vector"], **kw)
        yc = cunet(xin, tt, inp["crossattn"], inp["vector"], control=list(ctrls), **kw)
    assert len(ctrls) == int(G64["n_ctrl"])
    assert rel(y, G64["unet_out_f32"]) < RTOL
    assert rel(yc, G64["cunet_out_f32"]) < RTOL
    for i, c in enumerate(ctrls):
        assert rel(c, G64[f"ctrl_{i}_f32"]) < RTOL, i
    # the reference's own reduced-precision error is what the bf16 GPU path is budgeted against: it must be a real number
    assert 1e-3 < rel(G64["unet_out_bf16ac"], G64["unet_out_f32"]) < 0.1


def test_production_width_config_matches_reference_golden(golden_dir):
    """The production channel widths (model_channels 320 -> 320 / 640 channels, num_head_channels 64) on a 16x16 latent: CPU fp32
    restatement vs the reference's fp32 outputs (tests/golden/sgm_c320.npz, tools/gen_golden_sgm_c320.py). The GPU suite runs the same
    nets in bf16 / f16 through the implicit-GEMM convolutions, the token-major VideoResBlock and the MFMA temporal attention and holds
    them to the reference's own autocast error recorded in the same fixture."""
    G = np.load(os.path.join(golden_dir, "sgm_c320.npz"))
    unet = instantiate_from_config({"target": "sgm.modules.diffusionmodules.video_model.VideoUNet", "params": H.SMALL_UNET320}).eval()
    cunet = instantiate_from_config({"target": "models.csvd.ControlledVideoUNet", "params": H.SMALL_UNET320}).eval()
    cnet = instantiate_from_config({"target": "models.csvd.ControlNet", "params": H.SMALL_CTRL320}).eval()
    unet.load_state_dict(H.seeded_state_dict(unet, 51), strict=True)
    cunet.load_state_dict(H.seeded_state_dict(cunet, 51), strict=True)
    cnet.load_state_dict(H.seeded_state_dict(cnet, 52), strict=True)
    inp = H.seeded_inputs(53, hw=H.LATENT_HW320, cfg=H.SMALL_UNET320)
    inp["image_only_indicator"][0, 1] = 1.0
    kw = dict(num_video_frames=H.T_FRAMES, image_only_indicator=inp["image_only_indicator"])
    xin = torch.cat([inp["x"], inp["concat"]], 1)
    tt = 0.25 * inp["sigma"].log()
    probes = {}
    for name in H.C320_PROBES:                  # intermediate block outputs, subsampled like the fixture
        unet.get_submodule(name).register_forward_hook(
            lambda m, i, o, name=name: probes.__setitem__(name, o.detach().float()[:, ::4, ::2, ::2].contiguous()))
    with torch.no_grad():
        y = unet(xin, tt, inp["crossattn"], inp["vector"], **kw)
        ctrls = cnet(xin, inp["control_hint"], tt, inp["crossattn"], inp["vector"], **kw)
        yc = cunet(xin, tt, inp["crossattn"], inp["vector"], control=list(ctrls), **kw)
    assert len(ctrls) == int(G["n_ctrl"])
    for name in H.C320_PROBES:
        assert rel(probes[name], G[f"probe_{name}_f32"]) < RTOL, name
    assert rel(y, G["unet_out_f32"]) < RTOL
    assert rel(yc, G["cunet_out_f32"]) < RTOL
    assert rel(ctrls[-1], G["ctrl_last_f32"]) < RTOL
    assert 1e-3 < rel(G["unet_out_bf16ac"], G["unet_out_f32"]) < 0.1 and 1e-4 < rel(G["cunet_out_f16ac"], G["cunet_out_f32"]) < 0.1


def test_two_stream_gate_is_off_without_a_pinned_gemm_set(monkeypatch):
    """engine.two_streams_active (round 6): the ControlNet runs beside the UNet encoder only when the library GEMM set is pinned in the
    process (TunableOp enabled in look-up-only mode with the shipped file's validators) or when forced; a process that never pinned it —
    this one: no GPU, TunableOp untouched — runs one stream. MVI_SVD_TWO_STREAMS forces either way."""
    from multiview_inpaint_amd.svd import engine as E
    monkeypatch.setattr(E, "_pinned", None)
    monkeypatch.setattr(E, "TWO_STREAMS", None)
    assert E.gemm_set_pinned() is False and E.two_streams_active() is False
    monkeypatch.setattr(E, "TWO_STREAMS", True)
    assert E.two_streams_active() is True
    monkeypatch.setattr(E, "TWO_STREAMS", False)
    monkeypatch.setattr(E, "_pinned", True)
    assert E.two_streams_active() is False
    monkeypatch.setattr(E, "TWO_STREAMS", None)
    assert E.two_streams_active() is True                      # gate open: the default follows it
