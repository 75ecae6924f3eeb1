"""Micro-benchmark of the HIP attention kernels at the SVD shapes, random data.
Usage (GPU box): python tools/bench_attention.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multiview_inpaint_amd.svd import hip_ops  # noqa: E402

SHAPES = [(28, 5, 9216, 9216, 64), (28, 10, 2304, 2304, 64), (28, 20, 576, 576, 64), (28, 20, 144, 144, 64),
          (18432, 5, 14, 14, 64)]
QUICK = "--quick" in sys.argv
for dtype in ((torch.bfloat16,) if QUICK else (torch.bfloat16, torch.float16)):
    for B, H, Sq, Sk, D in (SHAPES[:2] if QUICK else SHAPES):
        g = torch.Generator(device="cuda").manual_seed(0)
        q, k, v = (torch.randn(B, s, H * D, device="cuda", generator=g).to(dtype) for s in (Sq, Sk, Sk))
        for _ in range(2):
            hip_ops.attention(q, k, v, H)
        torch.cuda.synchronize()
        n = 5 if Sq > 4000 else 20
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            hip_ops.attention(q, k, v, H)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / n
        fl = 4.0 * B * H * Sq * Sk * D
        print(f"{str(dtype):16s} B{B} H{H} S{Sq}x{Sk}: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s  kind={hip_ops.attention_kernel_kind(Sq, Sk, D, dtype)}", flush=True)
