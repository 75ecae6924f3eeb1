"""Times the first-stage decode (and encode) of the SVD pipeline at the bench size: 14 frames, 72x128 latents ->
576x1024 RGB, fp32 (the reference disables autocast for the first stage), seeded random weights, on cuda:0.
Prints one JSON object with the per-op HIP kernel times (hipEvents on the launch stream) beside the total.

    python tools/bench_vae.py [--frames 14] [--h 72] [--w 128] [--iters 3] [--dtype fp32|bf16]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FULL = dict(attn_type="vanilla", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
            ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=14)
    ap.add_argument("--h", type=int, default=72)
    ap.add_argument("--w", type=int, default=128)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--encode", action="store_true")
    ap.add_argument("--search", action="store_true", help="let MIOpen time its solvers in the warm-up (minutes in fp32)")
    a = ap.parse_args()
    from multiview_inpaint_amd.svd import hip_ops, vae
    import svd_helpers as H
    dev = "cuda:0"
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    torch.backends.cudnn.benchmark = a.search
    eng = vae.AutoencodingEngine(encoder_config=vae.Encoder(**FULL),
                                 decoder_config=vae.VideoDecoder(**FULL, video_kernel_size=[3, 1, 1])).eval()
    eng.decoder.load_state_dict(H.seeded_state_dict(eng.decoder, 42))
    eng.encoder.load_state_dict(H.seeded_state_dict(eng.encoder, 41))
    eng = eng.to(dev)                                            # the model stays fp32: --dtype bf16 is decode_first_stage's opt-in copy
    g = torch.Generator().manual_seed(0)
    z = (torch.randn(a.frames, 4, a.h, a.w, generator=g) * 0.18215).to(dev)
    out = {"workload": f"first-stage decode, {a.frames} frames, latent {a.h}x{a.w} -> {8 * a.h}x{8 * a.w}, {a.dtype}, "
                       "seeded random weights", "iters": a.iters}

    def run(fn, arg):
        with torch.no_grad():
            print("warm-up ...", file=sys.stderr, flush=True)
            t0 = time.perf_counter()
            y = fn(arg)                                           # warm-up (MIOpen solver search with --search)
            torch.cuda.synchronize()
            print(f"warm-up done in {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
            hip_ops.PROFILE = []
            t0 = time.perf_counter()
            for _ in range(a.iters):
                y = fn(arg)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / a.iters
        prof = hip_ops.profile_summary()
        hip_ops.PROFILE = None
        ops = {k: {"calls": c // a.iters, "ms": round(t / a.iters, 3), "GBs_or_GFLOPs": round(w / t / 1e6, 1) if t else None}
               for k, (c, t, w) in prof.items()}
        return y, ms, ops
    y, ms, ops = run(lambda t: vae.decode_first_stage(eng, t, dtype=dt), z)
    out["decode"] = {"ms": round(ms, 2), "frames_per_s": round(a.frames / ms * 1e3, 2), "finite": bool(torch.isfinite(y).all()),
                     "out_shape": list(y.shape), "hip_ops": ops, "hip_ops_ms": round(sum(v["ms"] for v in ops.values()), 2)}
    if a.encode:
        x = torch.rand(a.frames, 3, 8 * a.h, 8 * a.w, generator=g).to(dev) * 2 - 1
        zz, ms, ops = run(lambda t: vae.encode_first_stage(eng, t), x)
        out["encode"] = {"ms": round(ms, 2), "finite": bool(torch.isfinite(zz).all()), "hip_ops": ops}
    out["peak_mem_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
