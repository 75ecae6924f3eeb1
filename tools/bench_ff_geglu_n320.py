"""GEGLU projection at K = 640 / 1280 (csrc/linear_n320.hip, kGeglu: mvi_ff_geglu_n320) against the library GEMM + geglu kernel at
the level-1 / level-2 FeedForward shapes of the 14 x 576x1024 step ([64512, 640] x [640, 2 x 2560], [16128, 1280] x [1280, 2 x 5120],
bf16; the temporal twins have the same shapes): results vs fp64 on a row sample, time of both forms interleaved in one process.
The library GEMMs run through the shipped TunableOp selections (as the step does). Run on the GPU box: python tools/bench_ff_geglu_n320.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multiview_inpaint_amd.svd import bench_svd, hip_ops  # noqa: E402

bench_svd.enable_gemm_tuning()
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
for rows, K, inner, dtype in [(64512, 640, 2560, torch.bfloat16), (16128, 1280, 5120, torch.bfloat16), (64512, 640, 2560, torch.float16),
                              (258048, 640, 2560, torch.bfloat16),
                              # round 6: the level-0 shape (K = 320) through this form, against csrc/ff_geglu.hip's K = 320 kernel
                              (258048, 320, 1280, torch.bfloat16), (258048, 320, 1280, torch.float16)]:
    x = (torch.randn(rows, K, device=dev, generator=g) * 1.2).to(dtype)
    w = (torch.randn(2 * inner, K, device=dev, generator=g) * K ** -0.5).to(dtype)
    b = (torch.randn(2 * inner, device=dev, generator=g) * 0.3).to(dtype)
    y = hip_ops.ff_geglu_n320(x, w, b)
    unf = hip_ops.geglu(F.linear(x, w, b))
    idx = torch.randint(0, rows, (512,), device=dev, generator=g)
    idx[0], idx[1] = 0, rows - 1
    h = F.linear(x[idx].double(), w.double(), b.double())
    ref = h[:, :inner] * F.gelu(h[:, inner:])
    sc = float(ref.abs().max())
    e_f = float((y[idx].double() - ref).abs().max()) / sc
    e_u = float((unf[idx].double() - ref).abs().max()) / sc
    fns = (lambda: hip_ops.ff_geglu_n320(x, w, b), lambda: hip_ops.geglu(F.linear(x, w, b)), lambda: F.linear(x, w, b))
    if K == 320:                                             # the second column is the K = 320 kernel of csrc/ff_geglu.hip instead of the library
        fns = (fns[0], lambda: hip_ops.ff_geglu(x, w, b), fns[2])
    ts = [[], [], []]
    for rnd in range(5):                                     # interleaved rounds: both forms see the same clock history
        for k, fn in enumerate(fns):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                fn()
            e.record()
            torch.cuda.synchronize()
            ts[k].append(a.elapsed_time(e) / 10)
    med = [sorted(t)[len(t) // 2] for t in ts]
    fl = 4.0 * rows * K * inner
    print(f"rows {rows} K {K} inner {inner} {str(dtype)[6:]}: fused {med[0] * 1e3:7.1f} us ({fl / med[0] * 1e-9:6.1f} TFLOP/s; min {min(ts[0]) * 1e3:.1f})  "
          f"{'ff_geglu_k320 kernel' if K == 320 else 'library GEMM + geglu'} {med[1] * 1e3:7.1f} us (GEMM alone {med[2] * 1e3:.1f}, {fl / med[2] * 1e-9:.1f} TFLOP/s);  "
          f"max err / max |ref|: fused {e_f:.2e}  unfused {e_u:.2e}", flush=True)
