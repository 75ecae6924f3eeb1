"""Fused GEGLU projection (csrc/ff_geglu.hip) against library GEMM + geglu kernel at the level-0 FeedForward shape of the
14 x 576x1024 step ([258048, 320] x [320, 2 x 1280], bf16): results vs fp64 on a row sample, time of both forms.
Run on the GPU box:  python tools/bench_ff_geglu.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multiview_inpaint_amd.svd import hip_ops  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
for rows, K, inner, dtype in [(258048, 320, 1280, torch.bfloat16), (1000, 320, 1280, torch.bfloat16), (258048, 320, 1280, torch.float16)]:
    x = (torch.randn(rows, K, device=dev, generator=g) * 1.2).to(dtype)
    w = (torch.randn(2 * inner, K, device=dev, generator=g) * K ** -0.5).to(dtype)
    b = (torch.randn(2 * inner, device=dev, generator=g) * 0.3).to(dtype)
    y = hip_ops.ff_geglu(x, w, b)
    unf = hip_ops.geglu(F.linear(x, w, b))
    idx = torch.randint(0, rows, (512,), device=dev, generator=g)
    idx[0], idx[1] = 0, rows - 1
    h = F.linear(x[idx].double(), w.double(), b.double())
    ref = h[:, :inner] * F.gelu(h[:, inner:])
    sc = float(ref.abs().max())
    e_f = float((y[idx].double() - ref).abs().max()) / sc
    e_u = float((unf[idx].double() - ref).abs().max()) / sc
    ts = []
    for fn in (lambda: hip_ops.ff_geglu(x, w, b), lambda: hip_ops.geglu(F.linear(x, w, b))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) / 10)
    fl = 4.0 * rows * K * inner
    print(f"rows {rows} K {K} inner {inner} {str(dtype)[6:]}: fused {ts[0] * 1e3:7.1f} us ({fl / ts[0] * 1e-9:6.1f} TFLOP/s)  GEMM + geglu {ts[1] * 1e3:7.1f} us;"
          f"  max err / max |ref|: fused {e_f:.2e}  unfused {e_u:.2e}", flush=True)

# the same kernel with the plain epilogue against the library GEMM: the bias-only level-0 projections
for rows, K, N, with_bias in [(258048, 320, 960, False), (258048, 320, 320, True), (64512, 320, 960, False)]:
    dtype = torch.bfloat16
    x = (torch.randn(rows, K, device=dev, generator=g) * 1.2).to(dtype)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(dtype)
    b = (torch.randn(N, device=dev, generator=g) * 0.3).to(dtype) if with_bias else None
    y = hip_ops.linear_k320(x, w, b)
    lib = F.linear(x, w, b)
    idx = torch.randint(0, rows, (512,), device=dev, generator=g)
    ref = F.linear(x[idx].double(), w.double(), None if b is None else b.double())
    sc = float(ref.abs().max())
    e_k, e_l = float((y[idx].double() - ref).abs().max()) / sc, float((lib[idx].double() - ref).abs().max()) / sc
    ts = []
    for fn in (lambda: hip_ops.linear_k320(x, w, b), lambda: F.linear(x, w, b)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) / 10)
    byt = rows * (K + N) * 2
    print(f"linear rows {rows} {K} -> {N} bias {int(with_bias)}: k320 kernel {ts[0] * 1e3:7.1f} us ({byt / ts[0] * 1e-6:6.0f} GB/s)  library {ts[1] * 1e3:7.1f} us;"
          f"  max err / max |ref|: {e_k:.2e} vs {e_l:.2e}", flush=True)

# the projection INTO 320 channels with a long contraction (csrc/linear_n320.hip): FeedForward.net[2] at level 0
for rows, K, with_bias in [(258048, 1280, True), (64512, 1280, True)]:
    N = 320
    dtype = torch.bfloat16
    x = (torch.randn(rows, K, device=dev, generator=g) * 1.2).to(dtype)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(dtype)
    b = (torch.randn(N, device=dev, generator=g) * 0.3).to(dtype) if with_bias else None
    y = hip_ops.linear_n320(x, w, b)
    lib = F.linear(x, w, b)
    idx = torch.randint(0, rows, (512,), device=dev, generator=g)
    ref = F.linear(x[idx].double(), w.double(), None if b is None else b.double())
    sc = float(ref.abs().max())
    e_k, e_l = float((y[idx].double() - ref).abs().max()) / sc, float((lib[idx].double() - ref).abs().max()) / sc
    ts = []
    for fn in (lambda: hip_ops.linear_n320(x, w, b), lambda: F.linear(x, w, b)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) / 10)
    fl = 2.0 * rows * K * N
    print(f"linear rows {rows} {K} -> {N} bias {int(with_bias)}: n320 kernel {ts[0] * 1e3:7.1f} us ({fl / ts[0] * 1e-9:6.1f} TFLOP/s)  library {ts[1] * 1e3:7.1f} us "
          f"({fl / ts[1] * 1e-9:6.1f} TFLOP/s);  max err / max |ref|: {e_k:.2e} vs {e_l:.2e}", flush=True)
