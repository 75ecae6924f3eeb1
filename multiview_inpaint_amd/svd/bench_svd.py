"""SVD denoise-step benchmark (BASELINE.json metric, second half: "SVD 14-frame 576x1024 denoise
steps/s"). One step = one CFG-doubled (batch 28) evaluation of ControlNet + ControlledVideoUNet
through Denoiser.forward, bf16 autocast, synthetic tensors of the shapes in SURVEY.md §8d, seeded
N(0, 0.02) weights (no pretrained weights exist offline). Configuration = the reference YAML
svd_inpaint1/configs/test/svd_f_est_ctrl_simp1.yaml:19-61."""
import os
import sys
import time

import torch

SVD_UNET = dict(in_channels=8, out_channels=4, model_channels=320, channel_mult=[1, 2, 4, 4], num_res_blocks=2,
                attention_resolutions=[4, 2, 1], num_head_channels=64, transformer_depth=1, context_dim=1024,
                adm_in_channels=768, num_classes="sequential", use_linear_in_transformer=True, extra_ff_mix_layer=True,
                use_spatial_context=True, merge_strategy="learned_with_images", video_kernel_size=[3, 1, 1],
                use_checkpoint=True, spatial_transformer_attn_type="softmax-xformers")
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16
HBM_PEAK_GBS = 8000.0


def build(device, seed=0, with_control=True, dtype=torch.float32):
    """Full-size networks materialised directly on `device` (meta construction, seeded N(0, 0.02))."""
    from .engine import SVDInpaintEngine
    from .schedule import Denoiser
    from .unet import ControlledVideoUNet, ControlNet
    with torch.device("meta"):
        unet = ControlledVideoUNet(**SVD_UNET)
        cnet = ControlNet(hint_channels=7, **{k: v for k, v in SVD_UNET.items() if k != "out_channels"}) if with_control else None
    g = torch.Generator(device=device).manual_seed(seed)
    for m in (unet, cnet):
        if m is None:
            continue
        m.to_empty(device=device)
        with torch.no_grad():
            for name, p in list(m.named_parameters()) + list(m.named_buffers()):
                if name.endswith("mix_factor"):
                    p.fill_(0.5)
                elif p.ndim == 1 and name.endswith("weight") and ("norm" in name or "layers.0" in name or name.startswith("out.0")):
                    p.normal_(1.0, 0.02, generator=g)
                else:
                    p.normal_(0.0, 0.02, generator=g)
        m.eval().to(dtype)
    den = Denoiser({"target": "multiview_inpaint_amd.svd.schedule.VScalingWithEDMcNoise"})
    return SVDInpaintEngine(unet, cnet, den)


def inputs(device, T=14, h=72, w=128, seed=0, cfg_doubled=True):
    g = torch.Generator(device=device).manual_seed(seed)
    B = (2 if cfg_doubled else 1) * T
    r = lambda *s: torch.randn(*s, device=device, generator=g)
    cond = dict(crossattn=r(B, 1, 1024), vector=r(B, 768), concat=r(B, 4, h, w),
                control_hint=torch.rand(B, 7, 8 * h, 8 * w, device=device, generator=g))
    return r(B, 4, h, w), cond, torch.zeros(B // T if not cfg_doubled else B // (2 * T), T, device=device)


MIOPEN_USERDB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_userdb")


def use_shipped_miopen_db():
    """Points MIOpen at a private copy of the find-db recorded on the MI355X box for this step's convolution shapes
    (multiview_inpaint_amd/svd/miopen_userdb/*.ufdb.txt / *.udb.txt: problem -> fastest solver, written by MIOpen itself
    during a run with an empty MIOPEN_USER_DB_PATH). With it the solver search of the warm-up step is a look-up (bench
    start-up 60 -> 44 s) and every run uses the same convolution kernels instead of whatever a noisy search picked
    (observed 205 ... 225 ms per step across runs). Must be called before the process touches MIOpen; does nothing when
    MIOPEN_USER_DB_PATH is already set or MVI_SVD_MIOPEN_DB=0. The copy keeps MIOpen's own updates out of the repository."""
    if os.environ.get("MVI_SVD_MIOPEN_DB", "1") == "0" or "MIOPEN_USER_DB_PATH" in os.environ or not os.path.isdir(MIOPEN_USERDB):
        return os.environ.get("MIOPEN_USER_DB_PATH")
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="mvi_miopen_")
    for f in os.listdir(MIOPEN_USERDB):
        shutil.copy(os.path.join(MIOPEN_USERDB, f), os.path.join(d, f))
    os.environ["MIOPEN_USER_DB_PATH"] = d
    return d


TUNED_GEMMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")


def enable_gemm_tuning(tune_ms=int(os.environ.get("MVI_SVD_GEMM_TUNING_MS", "0"))):
    """PyTorch-ROCm's TunableOp for the library GEMMs behind the Linear layers (hipBLASLt / rocBLAS, ~40 shapes per step):
    the GEMM counterpart of MIOpen's solver search for the convolutions. Tuning inside the benchmark proved unreliable with
    budgets that fit a warm-up step (20 ms per shape: from 206 to 285 ms per step depending on the run, against 212
    untuned), so the solutions are selected OFFLINE with a generous budget (tools/tune_svd_gemms.sh on the MI355X box ->
    multiview_inpaint_amd/svd/tunableop_gfx950.csv: op signature -> library solution index, validated against the
    PyTorch / hipBLASLt / rocBLAS versions recorded in its header) and only LOOKED UP here: no tuning at run time.
    MVI_SVD_GEMM_TUNING_MS > 0 re-tunes in place instead (what the tuning script does); MVI_SVD_TUNED_GEMMS=0 runs the
    library defaults."""
    if os.environ.get("MVI_SVD_TUNED_GEMMS", "1") == "0":
        return False
    try:
        import torch.cuda.tunable as tn
        if tune_ms > 0:
            tn.enable(True)
            tn.tuning_enable(True)
            tn.set_max_tuning_duration(tune_ms)
            tn.set_max_tuning_iterations(int(os.environ.get("MVI_SVD_GEMM_TUNING_ITERS", "30")))
            tn.set_filename(os.environ.get("MVI_SVD_GEMM_TUNING_OUT", os.path.join(os.environ.get("TMPDIR", "/tmp"), "mvi_tunableop.csv")))
            return True
        if not os.path.exists(TUNED_GEMMS):
            return False
        tn.enable(True)
        tn.tuning_enable(False)                              # look-up only
        # a private copy, as for MIOpen: TunableOp may rewrite its results file, and under torch.distributed.run every
        # rank would do that to the same in-tree file
        import shutil
        import tempfile
        private = os.path.join(tempfile.mkdtemp(prefix="mvi_tunableop_"), os.path.basename(TUNED_GEMMS))
        shutil.copy(TUNED_GEMMS, private)
        tn.set_filename(private)
        return True
    except Exception as e:                                  # an older PyTorch: run untuned
        print(f"[mvi] TunableOp unavailable ({e}); GEMMs run with the library's default solutions", file=sys.stderr)
        return False


def library_selection_status(device):
    """Did the shipped solver / solution tables actually take effect in THIS process? Both are keyed to library builds and a
    mismatch is otherwise silent (the step just runs ~20 % slower on whatever the libraries pick):
      * TunableOp: the csv's `Validator` lines against torch.cuda.tunable.get_validators() of the running stack;
      * MIOpen: MIOpen names its user-db files after its own version — after a step, a db file in MIOPEN_USER_DB_PATH that is
        not one of the shipped ones means the installed MIOpen looked its problems up under another name and missed.
    Returns (dict for the bench line, list of warning strings); bench_svd prints the warnings to stderr."""
    status, warns = {}, []
    try:
        import torch.cuda.tunable as tn
        want = {}
        with open(TUNED_GEMMS) as fh:
            for line in fh:
                if line.startswith("Validator,"):
                    _, k, v = line.rstrip("\n").split(",", 2)
                    want[k] = v
        have = {str(k): str(v) for k, v in (tn.get_validators() or ())} if tn.is_enabled() else {}
        bad = {k: (v, have.get(k)) for k, v in want.items() if have and have.get(k) != v}
        status["tunableop_file_matches_stack"] = bool(have) and not bad
        if bad:
            warns.append(f"TunableOp file {os.path.basename(TUNED_GEMMS)} was recorded on another stack {bad}: PyTorch ignores it and "
                         "the GEMMs run with the library's default solutions (re-record with tools/tune_svd_gemms.sh)")
    except Exception as e:                                   # TunableOp absent: nothing selected, nothing to mismatch
        status["tunableop_file_matches_stack"] = None
        warns.append(f"TunableOp status unavailable ({e})")
    d = os.environ.get("MIOPEN_USER_DB_PATH")
    if d and os.path.isdir(d) and os.path.isdir(MIOPEN_USERDB):
        shipped = set(os.listdir(MIOPEN_USERDB))
        extra = sorted(f for f in os.listdir(d) if f not in shipped and f.endswith(("udb.txt", "ufdb.txt")))
        status["miopen_db_matches_library"] = not extra
        if extra:
            warns.append(f"the shipped MIOpen find-db ({sorted(shipped)}) is keyed to another MIOpen build: this one wrote {extra}; "
                         "its convolutions ran on immediate-mode fallbacks (re-record with tools/regen_miopen_db.sh)")
    else:
        status["miopen_db_matches_library"] = None
    return status, warns


def run_sample_loop(eng, device, num_steps=25, T=14, h=72, w=128, weights="bf16"):
    """One whole sample through SVDInpaintEngine.sample(): EulerEDMSampler (sampling.py:110-131) with per-frame linear
    guidance over c / uc (CFG batch 2T) — the loop the per-step metric is a slice of. Here the step-invariant work is
    done once per sample (ControlNet hint stem, doubled conditioning), which the per-step metric never skips."""
    from .schedule import EulerEDMSampler
    _, c, ind = inputs(device, T, h, w, cfg_doubled=False)
    if weights in _HALF:
        c = {k: v.to(_HALF[weights]) for k, v in c.items()}
    uc = {k: (v if k == "control_hint" else torch.zeros_like(v)) for k, v in c.items()}
    sch = "multiview_inpaint_amd.svd.schedule."
    eng.sampler = EulerEDMSampler(num_steps=num_steps, device=device,
                                  discretization_config={"target": sch + "EDMDiscretization", "params": {"sigma_max": 700.0}},
                                  guider_config={"target": sch + "LinearPredictionGuider",
                                                 "params": {"max_scale": 2.5, "min_scale": 1.0, "num_frames": T,
                                                            "additional_cond_keys": ["control_hint"]}})
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(weights not in _HALF)):
        out = eng.sample(None, c, uc=uc, batch_size=T, shape=(4, h, w), num_video_frames=T, image_only_indicator=ind)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    return dict(steps=num_steps, seconds=round(dt, 3), ms_per_step=round(dt / num_steps * 1e3, 2),
                note="EulerEDMSampler + LinearPredictionGuider(1.0 -> 2.5) over the same networks and shapes; hint stem and "
                     "doubled conditioning evaluated once per sample",
                finite=bool(torch.isfinite(out).all()))


def run_first_stage_decode(device, T=14, h=72, w=128, iters=2):
    """What follows the 25 steps of a sample: the first-stage decode of its T latent frames to 8 h x 8 w RGB frames
    (sgm/models/diffusion.py:194-212; VideoDecoder of configs/test/svd_f_est_ctrl_simp1.yaml:131-159 at full width, seeded random
    weights), in the DEFAULT configuration — the reference's fp32 contract (disable_first_stage_autocast), since round 6 with the
    convolutions on the bf16 matrix pipe with split operands (svd/vae_split.py). Reported beside the denoise step, not part of it."""
    from . import hip_ops, vae
    full = dict(attn_type="vanilla", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
                num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    dec = vae.VideoDecoder(**full, video_kernel_size=[3, 1, 1]).eval()
    g = torch.Generator().manual_seed(42)
    with torch.no_grad():
        for k, p in sorted(dec.state_dict().items()):           # seeded like tests/svd_helpers.seeded_state_dict: unit-gain weights
            if p.dtype.is_floating_point:
                r = torch.randn(p.shape, generator=g)
                p.copy_(r if k.endswith("mix_factor") else r / p[0].numel() ** 0.5 if p.ndim >= 2 else (1.0 + 0.1 * r if k.endswith("weight") else 0.1 * r))
    eng = vae.AutoencodingEngine(encoder_config=torch.nn.Identity(), decoder_config=dec.to(device))
    z = (torch.randn(T, 4, h, w, generator=g) * 0.18215).to(device)
    torch.cuda.empty_cache()                                       # (the 26 GB of this decode come after a sample loop's worth of cached blocks)
    with torch.no_grad():
        y = vae.decode_first_stage(eng, z)                         # warm-up
        torch.cuda.synchronize(device)
        hip_ops.PROFILE = []
        each = []
        for _ in range(max(iters, 3)):                             # each decode timed on its own; the MEDIAN is reported, all are listed (one
            t0 = time.perf_counter()                               # bench run of round 6 reported 787 ms as the mean of two; 398 everywhere else)
            y = vae.decode_first_stage(eng, z)
            torch.cuda.synchronize(device)
            each.append(round((time.perf_counter() - t0) * 1e3, 1))
        ms = sorted(each)[len(each) // 2]
        kinds = sorted(set(k for k, *_ in hip_ops.PROFILE))
        hip_ops.PROFILE = None
    return dict(ms=round(ms, 1), each_ms=each, frames=T, out_shape=list(y.shape), finite=bool(torch.isfinite(y).all()),
                convolutions="split bf16 operands on the matrix pipe, fp32 accumulate (fp32 contract, 1e-4)" if "conv_split3" in kinds
                else "fp32 library path", hip_ops=kinds)


_HALF = {"bf16": torch.bfloat16, "f16": torch.float16}


def run_gpu(device, steps=2, warmup=1, T=14, h=72, w=128, with_control=True, weights="bf16", sample_steps=25):
    """weights="bf16" (bench.py) / "f16" (the reference's precision): parameters stored in that type, no autocast (nothing is
    re-cast per step; GroupNorm statistics, softmax and LayerNorm still accumulate in fp32 inside their kernels).
    weights="fp32": fp32 parameters under torch.autocast(bf16), the reference's mixed-precision recipe
    (it uses fp16 autocast, models/csvd.py:27-31)."""
    from . import hip_ops, ops
    from .schedule import EDMDiscretization
    ops.STRICT = True               # every op of the timed step runs its HIP kernel or the run fails (svd/ops.py)
    # let MIOpen time its convolution solvers during warm-up (+10 % on the 3x3 convolutions; with the shipped find-db a look-up)
    # MIOpen solver choice: the shipped find-db is consulted in immediate mode too — same kernels, same step time as with
    # cudnn.benchmark (174.97 vs 174.14 ms same-box) — while benchmark mode re-runs its search for every channels-last
    # problem in every process (first step 164 s). MVI_SVD_MIOPEN_FIND=1 turns the search back on (e.g. to record a new db).
    torch.backends.cudnn.benchmark = os.environ.get("MVI_SVD_MIOPEN_FIND", "0") == "1"
    use_shipped_miopen_db()
    tuned = enable_gemm_tuning()
    eng = build(device, with_control=with_control, dtype=_HALF.get(weights, torch.float32))
    x, cond, ind = inputs(device, T, h, w)
    if weights in _HALF:
        cond = {k: v.to(_HALF[weights]) for k, v in cond.items()}      # conditioning is computed once per sample
    if not with_control:
        cond.pop("control_hint")
    sig = EDMDiscretization(sigma_max=700.0)(25, device=device)
    kw = dict(num_video_frames=T, image_only_indicator=ind)

    def step(i):
        s = sig[i % 25].expand(x.shape[0])
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=(weights not in _HALF)):
            return eng.denoise(x, s, cond, **kw)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize(device)
    # timed region: no per-op events (an event record idles the GPU for several microseconds, and a step has ~480 HIP ops)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]     # one record per 200 ms step: no measurable idle
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        out = step(i)
        marks[i + 1].record()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    per_step = [round(marks[i].elapsed_time(marks[i + 1]), 2) for i in range(steps)]
    # separate instrumented pass of the same steps: every HIP op bracketed by events on its launch stream -> per-op table.
    # The ControlNet runs on the main stream here: with the two networks sharing the chip (engine.TWO_STREAMS, the timed
    # region above) an op's events would time its kernel PLUS whatever ran beside it.
    from . import engine as _engine
    two_streams, ts_setting = _engine.two_streams_active(), _engine.TWO_STREAMS
    _engine.TWO_STREAMS = False
    hip_ops.PROFILE = []
    for i in range(steps):
        step(i)
    torch.cuda.synchronize(device)
    prof = hip_ops.profile_summary()
    hip_ops.PROFILE = None
    _engine.TWO_STREAMS = ts_setting
    res = dict(steps_per_s=round(1.0 / dt, 4), ms_per_step=round(dt * 1e3, 2), frames=T, latent=[h, w], batch=int(x.shape[0]),
               step_ms=per_step, controlnet=with_control, gemm_tuning=bool(tuned), two_streams=bool(two_streams and with_control),
               dtype=(f"{weights} weights + activations, fp32 GroupNorm statistics / softmax / LayerNorm accumulation" if weights in _HALF
                      else "bf16 autocast over fp32 weights, fp32 GroupNorm statistics / softmax"),
               finite=bool(torch.isfinite(out).all()))
    ops = {}
    for kind, (calls, ms, work) in prof.items():
        per_step_ms = ms / steps
        if kind in ("attention_mfma", "attention_rowtile", "ff_geglu", "ff_geglu_n320", "linear_n320", "linear_n320_ln", "conv3x3_n320", "conv3t_n320"):   # MFMA kernels: work = FLOPs
            tf = work / (ms * 1e-3) / 1e12
            ops[kind] = dict(calls_per_step=calls // steps, ms_per_step=round(per_step_ms, 3), TFLOPs=round(tf, 1),
                             frac_of_bf16_mfma_peak=round(tf / MFMA_BF16_PEAK_TFLOPS, 4))
        else:
            gbs = work / (ms * 1e-3) / 1e9
            ops[kind] = dict(calls_per_step=calls // steps, ms_per_step=round(per_step_ms, 3), GBs=round(gbs, 1),
                             frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 4))
    res["hip_ops"] = ops
    res["library_selection"], warns = library_selection_status(device)
    for w in warns:
        print(f"[mvi] WARNING: {w}", file=sys.stderr, flush=True)
    hip_ops.check_groupnorm_cluster(device)         # a benchmark number from a run with a timed-out GroupNorm wait is no number
    if sample_steps and with_control:
        res["sample_loop"] = run_sample_loop(eng, device, sample_steps, T, h, w, weights)
        del eng
        torch.cuda.empty_cache()
        try:
            res["first_stage_decode"] = run_first_stage_decode(device, T, h, w)
        except Exception as e:                                   # reported beside the step: never takes the step's number down with it
            res["first_stage_decode"] = {"error": repr(e)[:300]}
    return res


def run_cpu_baseline(threads=None):
    """BASELINE.json configs[0]: one Denoiser.forward of the full-size VideoUNet, 1 frame 256x256
    (latent 32x32), fp32, CPU PyTorch — this package's own modules on CPU tensors (the Python
    reference cannot travel to the GPU box; parity with it is pinned by tests/test_sgm_cpu.py)."""
    import os
    dev = torch.device("cpu")
    t0 = time.perf_counter()
    if torch.cuda.is_available():
        eng = build(torch.device("cuda"), with_control=False).to(dev)   # draw the 1.5 B weights on the GPU, run on CPU
    else:
        eng = build(dev, with_control=False)
    t_build = time.perf_counter() - t0
    x, cond, ind = inputs(dev, T=1, h=32, w=32, cfg_doubled=False)
    cond.pop("control_hint")
    s = torch.full((1,), 1.5)
    times = []
    with torch.no_grad():
        for _ in range(2):
            t1 = time.perf_counter()
            eng.denoise(x, s, cond, num_video_frames=1, image_only_indicator=ind)
            times.append(time.perf_counter() - t1)
    return dict(value=round(1.0 / times[-1], 4), unit="denoise steps/s", cores=torch.get_num_threads(), kind="port",
                sample=f"2 x Denoiser.forward(VideoUNet 1.52B params), 1 frame 256x256 (latent 32x32), fp32, "
                       f"{torch.get_num_threads()} torch threads of {os.cpu_count()} host cores; first call {times[0]:.1f} s, "
                       f"second {times[1]:.1f} s, model build {t_build:.1f} s")


if __name__ == "__main__":
    # one JSON line with run_gpu()'s result: the SVD leg of bench.py runs in a child process of its own (bench.py says why)
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sample-steps", type=int, default=25)
    ap.add_argument("--weights", choices=["bf16", "f16", "fp32"], default="bf16")
    a = ap.parse_args()
    use_shipped_miopen_db()
    res = run_gpu(torch.device("cuda", 0), steps=a.steps, warmup=a.warmup, sample_steps=a.sample_steps, weights=a.weights)
    print(json.dumps(res), flush=True)

