#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json metric).

A step = one forward + backward pass of the Gaussian-splat rasterizer-with-depth over one camera
view: 1.5 M synthetic Gaussians (SURVEY.md §8d recipe, seed 0), 1920x1080, sh_degree 3, upstream
gradient dL/dcolor ~ N(0,1). Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...`: one process per GPU, every rank renders a DIFFERENT view of the replicated scene
(weak scaling: one view per GPU per step) and the Gaussian gradients are summed with one RCCL
all-reduce of a flat bucket per step. value = pixels of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector)
# useful fp32 operations per evaluated (pixel, Gaussian) pair, counted from the kernels' arithmetic (csrc/raster_render.hip):
# forward: dx, dy 2, power 7, exp 1, alpha 2, test_T 2, colour + depth weights 8 = 22; backward: the forward's alpha / T replay
# 14, dL/dalpha incl. the background term 14, nine raw moments 18, colour accumulation 12, T / accumulators 6 = 64
PAIR_FLOPS = {"render_forward": 22, "render_backward": 64}


def algorithmic_bytes(N, V, D, M, P, T, n_pass=1):
    """SURVEY.md §8d closed form, per stage (bytes of compulsory traffic)."""
    fwd = dict(
        preprocess_forward=N * (44 + 12 * M) + N * 8 + V * 67,
        scan_block_sums=N * 8,
        duplicate_keys=V * 20 + D * 12,
        radix_sort=D * 24 * n_pass,
        tile_ranges=D * 8 + T * 8,
        render_forward=T * 8 + D * 44 + P * 24,
    )
    bwd = dict(
        render_backward=P * 20 + D * 40 + V * 44,
        preprocess_backward=N * (75 + 12 * M) + V * 44 + N * (40 + 12 * M),
    )
    return fwd, bwd


def view_camera(rank, W, H):
    """Rank r looks at the scene from a slightly different pose (yaw r*4 degrees about the scene centre)."""
    from multiview_inpaint_amd import synthetic as syn
    if rank == 0:
        return syn.make_camera(W, H, 50.0)
    a = math.radians(4.0 * rank)
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    centre = np.array([0.0, 0.0, 4.5])
    cam_pos = centre - R @ np.array([0.0, 0.0, 4.5])       # orbit at the same distance
    T = -R.T @ cam_pos
    return syn.make_camera(W, H, 50.0, R, T)


def cpu_baseline():
    """The oracle (a port: the reference has no CPU rasterizer) on a bounded sample: fwd+bwd of one 800x800 view at N=100k
    (BASELINE.json configs[1]) on the host cores this process may use (at most 32: the per-Gaussian and per-pixel loops of
    oracle/raster_oracle.c are split over threads for this measurement only; the binning stays on one thread)."""
    from multiview_inpaint_amd import synthetic as syn
    from oracle import raster_oracle as ro
    W = H = 800
    N = 100_000
    cam = syn.make_camera(W, H, 50.0)
    sc = syn.make_scene(N, cam, 3, seed=0)
    p = ro.make_params(N, 3, 16, W, H, cam["tanfovx"], cam["tanfovy"], 1.0, cam["viewmatrix"], cam["projmatrix"],
                       cam["campos"], np.zeros(3, np.float32))
    kw = dict(shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    g_img = np.random.default_rng(0).normal(size=(3, H, W)).astype(np.float32)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))
    ro.set_threads(cores)
    try:
        reps, t0 = 0, time.perf_counter()
        while True:
            f = ro.forward(p, sc["means3D"], sc["opacities"], **kw)
            ro.backward(p, f, g_img, sc["means3D"], **kw)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > 10.0 or reps >= 40:
                break
    finally:
        ro.set_threads(1)
    return dict(value=round(reps * W * H / dt / 1e6, 4), unit="Mpix/s", cores=cores, kind="port",
                sample=f"{reps} x (fwd+bwd, N=100k Gaussians, 800x800, sh_degree 3, D={f['num_rendered']}) "
                       f"with oracle/raster_oracle.c on {cores} OpenMP threads ({avail} of {os.cpu_count()} host cores available "
                       f"to this process), {dt:.1f} s")


SVD_CHILD_FAILURES = []          # what went wrong in a child that produced no result: kept in the JSON line, never swallowed


def _svd_child(two_streams, steps, timeout):
    """The SVD denoise-step benchmark (multiview_inpaint_amd/svd/bench_svd.py) in a child process; None if it failed or did not
    finish within `timeout` seconds (the child is then killed) — with the reason and the tail of the child's stderr appended to
    SVD_CHILD_FAILURES. two_streams None: the engine's own default (its gate, engine.gemm_set_pinned); False: forced to one stream."""
    import subprocess
    env = dict(os.environ)
    if two_streams is None:
        env.pop("MVI_SVD_TWO_STREAMS", None)
    else:
        env["MVI_SVD_TWO_STREAMS"] = "1" if two_streams else "0"
    what = "engine default" if two_streams is None else ("two streams" if two_streams else "one stream")

    def failed(why, err):
        tail = (err or "").strip().splitlines()[-12:]
        SVD_CHILD_FAILURES.append({"child": what, "why": why, "stderr_tail": tail})
        sys.stderr.write(f"[bench] SVD child ({what}) {why}\n" + "".join(f"    {ln}\n" for ln in tail))
        return None

    try:
        # MVI_BENCH_SVD_WEIGHTS=f16: the reference's own precision (fp16), +1.5 % step time on this chip; default bf16
        p = subprocess.run([sys.executable, "-m", "multiview_inpaint_amd.svd.bench_svd", "--steps", str(steps), "--warmup", "2",
                            "--weights", os.environ.get("MVI_BENCH_SVD_WEIGHTS", "bf16")],
                           cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        err = e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else e.stderr
        return failed(f"did not finish within {timeout} s and was killed", err)
    if p.returncode != 0:
        return failed(f"exited with code {p.returncode}", p.stderr)
    for line in reversed(p.stdout.strip().splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                return failed("printed a line that is not JSON", p.stderr)
    return failed("printed no JSON line", p.stderr)


def svd_leg(steps):
    """Second half of the BASELINE.json metric, measured in child processes started before this process initialises the GPU (a
    child may not be exec'd from a process that has). The HEADLINE is the engine's DEFAULT execution (round 6): the ControlNet on a
    side stream beside the UNet encoder exactly when the library GEMM set is pinned by the shipped TunableOp file in that process
    (engine.gemm_set_pinned — the condition under which the mode has completed every run since round 3; bench_svd pins it), one
    stream otherwise; `execution` says which ran. The forced one-stream step is measured as well and reported NEXT to the headline as
    `one_stream` (rounds 1 - 5 had it the other way round: one stream as the headline, `two_streams` beside it).
    MVI_BENCH_ONE_STREAM=0 skips the second child."""
    head = _svd_child(None, steps, 420)                  # (a normal run takes under a minute; the time-out is for a wedged child)
    if head is None:
        head = _svd_child(False, steps, 600)             # the default mode failed or hung (recorded in child_failures): one stream
        if head is None:
            return None
    head["execution"] = ("two streams: ControlNet beside the UNet encoder (the engine's default with the pinned GEMM set)" if head.get("two_streams")
                         else "one stream")
    if head.get("two_streams") and os.environ.get("MVI_BENCH_ONE_STREAM", "1") != "0":
        one = _svd_child(False, steps, 400)
        head["one_stream"] = ({k: one[k] for k in ("steps_per_s", "ms_per_step", "step_ms", "finite") if k in one}
                              if one is not None else "did not complete")
    return head


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=1_500_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--path", choices=["both", "raster", "svd"], default="both",
                    help="hot-path halves to run at N=1 (the SVD denoise loop is replicas-only across GPUs)")
    ap.add_argument("--svd-steps", type=int, default=5)
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    svd_result = None
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ)
    if world == 1 and args.gpus == 1 and args.path in ("both", "svd") and not under_profiler:
        # child processes, started BEFORE this process loads the HIP library or touches the GPU in any way (under rocprofv3 the
        # preloaded tool has already initialised the GPU: no children then, the SVD leg runs in this process further down)
        svd_result = svd_leg(args.svd_steps)
    if args.gpus > 1 and world == 1:
        # convenience: re-launch under torch.distributed.run as a child (never exec after GPU init)
        import subprocess
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    from multiview_inpaint_amd import _lib, raster as R, synthetic as syn
    from multiview_inpaint_amd import dist as mdist
    L = _lib.lib()                                        # fails loudly if the HIP library is missing
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # MVI_BENCH_DEVICE / MVI_BENCH_BACKEND: rehearsal of the multi-rank control flow on a ONE-GPU box (every rank on the same
    # device, gloo instead of RCCL, which refuses two ranks on one GPU) — never a performance number
    dev_index = int(os.environ.get("MVI_BENCH_DEVICE", local_rank))
    backend = os.environ.get("MVI_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    force_dist = os.environ.get("MVI_BENCH_FORCE_DIST") == "1"      # exercise the RCCL path with a 1-rank group
    if world > 1 or force_dist:
        import torch.distributed as td
        if force_dist and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            td.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        elif backend == "nccl":
            td.init_process_group("nccl", device_id=dev)
        else:
            td.init_process_group(backend)

    W, H, N, deg = args.width, args.height, args.gaussians, args.sh_degree
    M = (deg + 1) ** 2
    cam0 = syn.make_camera(W, H, 50.0)
    sc = syn.make_scene(N, cam0, deg, seed=0)             # identical on every rank (replicated scene)
    cam = view_camera(rank, W, H)
    t = {k: torch.tensor(v, device=dev) for k, v in sc.items() if k != "sh_degree"}
    rs = R.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
        bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=torch.tensor(cam["viewmatrix"], device=dev),
        projmatrix=torch.tensor(cam["projmatrix"], device=dev), sh_degree=deg,
        campos=torch.tensor(cam["campos"], device=dev), prefiltered=False)
    g_img = torch.randn(3, H, W, device=dev, generator=torch.Generator(dev).manual_seed(rank))
    kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    # exchange step: one flat all-reduce, or — when it moves fewer bytes (W < 2M) — the SH gradient in
    # factored form (all-gather of 3 floats per view and Gaussian + local rebuild) and an all-reduce of the other 11
    mode = os.environ.get("MVI_BENCH_EXCHANGE", "auto")
    n_ranges = int(os.environ.get("MVI_BENCH_RANGES", "1"))
    distributed = world > 1 or force_dist
    factored = distributed and (mode in ("factored", "compacted") or (mode == "auto" and mdist.FactoredGradExchange.pays(M, world)))
    # MVI_BENCH_RANGES=4: the other 11 floats are all-reduced in four Gaussian ranges, each started behind its own
    # chain-rule kernel (dist.RangedGradExchange). Not the default: on one rank (RCCL group of 1) the four smaller kernels,
    # the four collective calls and the assembly copy cost 0.19 ms per step (1.60 vs 1.41 ms), about what hiding three
    # quarters of a 66 MB all-reduce can return at 8 ranks — to be decided on an 8-GPU node, which this round never had
    ranged = factored and n_ranges > 1
    # "auto" (and "compacted") with more than one rank: only the rows inside the gradient support of at least one rank travel
    # (dist.CompactedGradExchange; the support of a view is 3 % of this scene's Gaussians, so the union over 8 views is a
    # fraction of the rows); it falls back to the full-size factored exchange when the union is above 80 %.
    # MVI_BENCH_EXCHANGE=factored / dense select the other forms.
    # (auto only where the factored form it is built on — and falls back to — moves fewer bytes than the dense bucket: W < 2M)
    compacted = distributed and M > 1 and (mode == "compacted" or (mode == "auto" and world > 1 and n_ranges <= 1 and
                                                                   mdist.FactoredGradExchange.pays(M, world)))
    bucket = (mdist.CompactedGradExchange(N, M, deg, dev) if compacted else
              mdist.RangedGradExchange(N, M, deg, dev, n_ranges=n_ranges) if ranged else
              mdist.FactoredGradExchange(N, M, deg, dev) if factored else mdist.GradBucket(N, M, dev))

    def step():
        color, radii, depth, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], prepare_backward=True, **kw)
        if compacted:
            R.rasterize_backward(rs, st, g_img, t["means3D"], out=bucket.views, sh_grad="factor", **kw)
            bucket.exchange_support(t["means3D"], rs.campos, st.tensor("grad_support", (N,), torch.uint8))
            return st, radii
        if ranged:
            R.rasterize_backward_ranged(rs, st, g_img, t["means3D"], t["shs"], t["scales"], t["rotations"], bucket)
            return st, radii
        if factored:
            # the all-gather of the colour factors starts right behind the render backward and runs under the chain rule
            R.rasterize_backward_split(rs, st, g_img, t["means3D"], t["shs"], t["scales"], t["rotations"], bucket.views,
                                       after_render=lambda: bucket.begin_gather(rs.campos))
            bucket.finish(t["means3D"])
            return st, radii
        R.rasterize_backward(rs, st, g_img, t["means3D"], out=bucket.views, sh_grad="dense", **kw)
        if distributed:
            bucket.all_reduce()
        return st, radii

    def barrier():
        if world > 1 or force_dist:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        st, radii = step()
    barrier()
    # Pass 1 (not timed): every stage bracketed by hipEvents on the launch stream -> the per-stage table. An event record
    # serialises the queue: ~10 us of idle GPU per bracketed stage boundary, 84 us per step with all 8 stages (measured:
    # tools/experiments/raster_gaps.sh), which is why the timed region below brackets ONLY the roofline kernel.
    L.mvi_raster_timing_enable(1)
    for _ in range(args.steps):
        st, radii = step()
    barrier()
    L.mvi_raster_timing_enable(0)
    ms_all = (C.c_float * 8)()
    calls_all = (C.c_int32 * 8)()
    _lib.check(L.mvi_raster_timing_read(ms_all, calls_all), "timing_read")
    dom_i = max(range(8), key=lambda i: ms_all[i])
    # Timed region: exactly K steps, the dominant kernel's launches timed live with events on its stream (roofline.achieved)
    L.mvi_raster_timing_enable_stages(1 << dom_i)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st, radii = step()
    barrier()
    dt = time.perf_counter() - t0
    L.mvi_raster_timing_enable(0)
    ms = (C.c_float * 8)()
    calls = (C.c_int32 * 8)()
    _lib.check(L.mvi_raster_timing_read(ms, calls), "timing_read")
    per_rank_ms = [round(dt / args.steps * 1e3, 4)]
    if world > 1 or force_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(tt) for _ in range(td.get_world_size())]
        td.all_gather(every, tt)                            # every rank's own wall time around the K steps
        per_rank_ms = [round(float(x.item()) / args.steps * 1e3, 4) for x in every]
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        D = st.D
        V = int((radii > 0).sum().item())
        T = ((W + 15) // 16) * ((H + 15) // 16)
        Ppix = W * H
        fwd_b, bwd_b = algorithmic_bytes(N, V, D, M, Ppix, T, n_pass=1)
        stage_bytes = {**fwd_b, **bwd_b}
        # The SURVEY §8d formula prices the per-Gaussian chain rule as dense: every Gaussian's parameters read, every gradient
        # written. Since round 3 only the gradient support G (Gaussians the render backward touched) is read and computed, and
        # the dense zero-fill of the outputs rides inside the render backward. The per-stage table uses that restated split
        # (so no stage reads above the HBM peak); `whole_step` keeps the §8d total for comparability across rounds and shows
        # the restated total next to it.
        G = int(st.tensor("grad_support", (N,), torch.uint8).sum().item()) if L.mvi_raster_backward_mode(-1) == 0 else N
        restated = dict(stage_bytes)
        # Since round 4 the SH rows (12 M bytes per Gaussian) are not read by the preprocess kernel: a colour is evaluated the
        # first time a tile stages the Gaussian, by the kernels of the render_forward stage (E Gaussians: their SH row, position
        # and the 17 bytes written back). The restated split moves those bytes; the §8d total stays what it was.
        rgbd = st.tensor("rgbd", (N, 4), torch.float32)
        E = int(((radii > 0) & ~(rgbd[:, :3] < 0).any(1)).sum().item()) if L.mvi_raster_color_mode(-1) == 1 else V
        if E < V:
            restated["preprocess_forward"] = fwd_b["preprocess_forward"] - N * 12 * M
            restated["render_forward"] = fwd_b["render_forward"] + E * (12 * M + 12 + 17)
        if G < N:
            restated["render_backward"] = bwd_b["render_backward"] + N * (52 + 12 * M)            # + zero-fill of the outputs
            restated["preprocess_backward"] = N + G * (129 + 12 * M) + G * (52 + 12 * M)           # flags + support rows in / out
        stages = {}
        for i in range(8):
            name = L.mvi_raster_stage_name(i).decode()
            if calls_all[i]:
                avg_ms = ms_all[i] / args.steps              # a stage may be bracketed more than once per step
                stages[name] = dict(ms=round(avg_ms, 4), GBs=round(restated[name] / avg_ms / 1e6, 1))
        dom = L.mvi_raster_stage_name(dom_i).decode()
        dom_ms = ms[dom_i] / args.steps                      # measured inside the timed region
        dom_gbs = round(stage_bytes[dom] / dom_ms / 1e6, 1)       # SURVEY §8d bytes of the kernel (not the restated ones)
        # HBM bytes per launch of the dominant stage from the committed PMC passes (FETCH_SIZE / WRITE_SIZE,
        # separate rocprofv3 --pmc runs, gfx950 correction applied: profiles/raster_traffic.json); null if absent
        # ... and only while they were measured on THIS build (digest of the rasterizer sources recorded in the file)
        traffic, issue = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "raster_traffic.json")) as fh:
                tj = json.load(fh)
            if tj.get("build") == _lib.raster_source_digest():
                traffic = tj["per_step_bytes_by_stage"].get(dom)
                issue = next((v for k, v in tj.get("issue_per_kernel", {}).items() if k.startswith(dom + "_kernel")), None)
        except (OSError, KeyError, ValueError):
            pass
        ms_step = dt / args.steps * 1e3
        total_bytes = sum(stage_bytes.values())
        ranges = st.tensor("ranges", (T, 2), torch.int32).long()
        lens = (ranges[:, 1] - ranges[:, 0]).float()
        nc = st.tensor("n_contrib", (H, W), torch.int32).float()
        ft = st.tensor("final_T", (H, W), torch.float32)
        pad_h, pad_w = (-H) % 16, (-W) % 16
        nct = torch.nn.functional.pad(nc, (0, pad_w, 0, pad_h)).reshape((H + pad_h) // 16, 16, (W + pad_w) // 16, 16)
        tile_max = nct.amax(dim=(1, 3))
        out = {
            "metric": "Mpix/s fwd+bwd @1.5M Gauss 1080p",
            "value": round(world * Ppix / (dt / args.steps) / 1e6, 2),
            "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            # one process per GPU; `value` uses the slowest rank's time. rccl_ranks = size of the RCCL group the exchange ran in
            "rccl_ranks": (td.get_world_size() if distributed and backend == "nccl" else 0), "per_rank_ms_per_step": per_rank_ms,
            **({"rehearsal": f"backend {backend}, every rank on cuda:{dev_index}: control flow only, not a performance number"}
               if backend != "nccl" or "MVI_BENCH_DEVICE" in os.environ else {}),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            # the shared library every kernel of this run came from (MVI_HIP_LIB redirects it for same-box A/B runs)
            "library": os.path.relpath(_lib.LIB_PATH, ROOT) + (" (MVI_HIP_LIB override)" if _lib.LIB_OVERRIDDEN else ""),
            "config": {"workload": f"rasterizer-with-depth fwd+bwd, one {W}x{H} view per GPU per step, "
                                   f"N={N} Gaussians, sh_degree {deg} (BASELINE.json configs[2] size, SURVEY.md §8d scene)",
                       "gaussians": N, "visible": V, "num_rendered_D": D, "tiles": T,
                       "tile_list_mean": round(float(lens.mean()), 1), "tile_list_max": int(lens.max()),
                       "n_contrib_mean": round(float(nc.mean()), 1), "tile_max_contrib_mean": round(float(tile_max.mean()), 1),
                       "pixels_saturated_frac": round(float((ft < 1e-3).float().mean()), 4),
                       "parallelism": f"views x{world}" + ((" + RCCL all-gather of SH colour factors + all-reduce of 11 floats/Gaussian"
                                                             + (f" in {n_ranges} ranges overlapped with the chain rule" if ranged else "")
                                                            if factored else " + RCCL all-reduce of the gradient bucket")
                                                           + (f" (support-compacted: union of the ranks' gradient supports {bucket.last_union_fraction:.3f} of the Gaussians, "
                                                              f"{'taken' if bucket.last_compacted else 'not taken'})" if compacted else "")
                                                           if distributed else "")},
            # achieved = algorithmic bytes of the dominant kernel's launch / its launch time, against the HBM peak.
            # The render kernels are not HBM-bound (most of their list gathers are L2 hits: `traffic` is BELOW the algorithmic
            # bytes), which is exactly what a low fraction says; what they ARE bound by rides along as information only:
            # `issue` (vector instructions issued per launch from the PMC passes, as a share of the launch's SIMD cycles — a
            # utilisation, not a roof: it rewards wasted instructions) and `compute` (useful fp32 work: evaluated (pixel,
            # Gaussian) pairs x FLOPs per pair against the 157.3 TFLOP/s vector peak).
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": dom_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(dom_gbs / HBM_PEAK_GBS, 5),
                         "traffic": traffic,
                         "launch_ms": round(dom_ms, 4),
                         "algorithmic_bytes_per_launch": int(stage_bytes[dom]),
                         "restated": {"bytes_per_launch": int(restated[dom]),
                                      "frac": round(restated[dom] / dom_ms / 1e6 / HBM_PEAK_GBS, 5),
                                      "note": "with the zero-fill of the dense gradient outputs that rides inside this kernel since round 3"},
                         "counters_build": _lib.raster_source_digest(),
                         "whole_step": {"algorithmic_bytes": int(total_bytes),
                                        "GBs": round(total_bytes / ms_step / 1e6, 1),
                                        "frac": round(total_bytes / ms_step / 1e6 / HBM_PEAK_GBS, 5),
                                        "formula": "SURVEY.md 8d, dense chain rule (comparable across rounds)",
                                        "restated_bytes": int(sum(restated.values())),
                                        "restated_frac": round(sum(restated.values()) / ms_step / 1e6 / HBM_PEAK_GBS, 5),
                                        "gradient_support": G, "colours_evaluated": E},
                         "compute": {"evaluated_pairs": int(nc.double().sum().item()),
                                     "flops_per_pair": PAIR_FLOPS.get(dom),
                                     "TFLOPs": (round(float(nc.double().sum().item()) * PAIR_FLOPS[dom] / dom_ms / 1e9, 2)
                                                if dom in PAIR_FLOPS else None),
                                     "peak": FP32_VECTOR_PEAK_TFLOPS,
                                     "frac": (round(float(nc.double().sum().item()) * PAIR_FLOPS[dom] / dom_ms / 1e9 / FP32_VECTOR_PEAK_TFLOPS, 4)
                                              if dom in PAIR_FLOPS else None),
                                     "note": "pairs = sum of n_contrib (list entries each pixel walks); FLOPs per pair counted "
                                             "from the kernel's arithmetic (DESIGN.md §4)"},
                         **({"issue": {"valu_wave_instructions_per_launch": int(issue["SQ_INSTS_VALU"]),
                                       "share_of_simd_cycles": issue["valu_issue_frac"]}} if issue else {})},
            "stages": stages,
            "stages_note": "per-stage times from a separate pass of K steps with every stage bracketed by hipEvents (each "
                           "bracketed boundary idles the GPU ~10 us); the timed region brackets only the roofline kernel. GB/s of "
                           "render_backward / preprocess_backward use the restated bytes (zero-fill inside the render backward, "
                           "chain rule on the gradient support only), preprocess_forward / render_forward the restated bytes of the "
                           "deferred SH colours (SH rows read for the colours_evaluated Gaussians only, by the render_forward stage: "
                           "mark_front + resolve_marked + render_forward kernels), every other stage the SURVEY 8d bytes",
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if world == 1 and args.path == "both":
            # SURVEY.md §8f rows either side of the rasterizer: one full training iteration, fused HIP ops vs PyTorch ops
            t = bucket = st = g_img = None
            torch.cuda.empty_cache()
            from multiview_inpaint_amd import bench_train
            out["train_iteration"] = bench_train.run_both(steps=20, warmup=5)
        if world == 1 and args.path in ("both", "svd"):
            # second half of the BASELINE.json metric: SVD 14-frame 576x1024 denoise steps/s
            t = bucket = st = g_img = None
            torch.cuda.empty_cache()
            from multiview_inpaint_amd.svd import bench_svd
            svd = svd_result                                  # measured in child processes at the start (svd_leg)
            if svd is None:
                # no child could run (e.g. this process was started under a profiler that had already initialised the GPU, so
                # the exec of a child is refused): the same measurement in this process, one stream
                bench_svd.use_shipped_miopen_db()
                svd = bench_svd.run_gpu(dev, steps=args.svd_steps, warmup=2)
                svd["execution"] = "one stream (in-process: the child process could not run)"
            if SVD_CHILD_FAILURES:
                svd["child_failures"] = SVD_CHILD_FAILURES
            svd["metric"] = "SVD 14-frame 576x1024 denoise steps/s (ControlNet + ControlledVideoUNet, CFG batch 28)"
            if not args.no_cpu_baseline:
                svd["cpu_baseline"] = bench_svd.run_cpu_baseline()
            out["svd"] = svd
            # HBM bytes per launch of the two hand-written contractions at their largest shapes of this step, from the committed
            # PMC passes (tools/pmc_svd_traffic.sh -> profiles/svd_traffic.json; gfx950 corrections applied there); null if absent
            svd_traffic = {}
            try:
                with open(os.path.join(ROOT, "profiles", "svd_traffic.json")) as fh:
                    svd_traffic = json.load(fh).get("kernels", {})
            except (OSError, ValueError):
                pass

            def _traffic(key):
                k = svd_traffic.get(key)
                return None if not k else {"hbm_bytes_per_launch": k["hbm_bytes_corrected"], "algorithmic_bytes_per_launch": k["algorithmic_bytes"],
                                           "launch": k["shape"], "l2_hit_rate": k.get("l2_hit_rate")}
            am = svd.get("hip_ops", {}).get("attention_mfma")
            if am:
                # the roofline of path B's hand-written contraction, flat at the top level (attention FLOPs /
                # summed launch time of the attention kernels / 2.5 PFLOP/s dense bf16)
                out["roofline_svd_attention"] = {
                    "bound": "mfma", "kernel": "attn_flash8m16_kernel (8 waves, v_mfma_f32_16x16x32; + attn_flash_kernel for S < 1024)",
                    "achieved": am["TFLOPs"], "peak": bench_svd.MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": am["frac_of_bf16_mfma_peak"], "calls_per_step": am["calls_per_step"], "ms_per_step": am["ms_per_step"],
                    "traffic": _traffic("attention")}
            cv = svd.get("hip_ops", {}).get("conv3x3_n320")
            if cv:
                # the other hand-written contraction of path B (round 3): the 3x3 convolutions as implicit GEMMs in
                # csrc/linear_n320.hip — 2 * pixels * 9 C_in * C_out FLOPs per call / summed launch time / 2.5 PFLOP/s
                out["roofline_svd_conv"] = {
                    "bound": "mfma", "kernel": "linear_n320_kernel<kConv> (3x3 convolutions, C_out = 320 g; K split at level 3)",
                    "achieved": cv["TFLOPs"], "peak": bench_svd.MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": cv["frac_of_bf16_mfma_peak"], "calls_per_step": cv["calls_per_step"], "ms_per_step": cv["ms_per_step"],
                    "traffic": _traffic("conv3x3_n320")}
            out["svd_steps_per_s"] = svd.get("steps_per_s")
    if world > 1 or force_dist:
        td.destroy_process_group()
    if rank == 0:
        C.CDLL(None).fflush(None)            # RCCL's banner sits in the C stdio buffer: flush it first so that the
        print(json.dumps(out), flush=True)   # JSON line is the LAST line of stdout


if __name__ == "__main__":
    main()
