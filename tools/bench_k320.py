"""Times csrc/ff_geglu.hip's two forms at the level-0 shapes of one denoise step (bf16, no checks: the parity tests do that).
For A/B builds: MVI_HIP_LIB=ab/<name>.so python tools/bench_k320.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_inpaint_amd.svd import hip_ops  # noqa: E402

torch.manual_seed(0)
dev = "cuda"
rows = 28 * 9216


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


x = torch.randn(rows, 320, device=dev, dtype=torch.bfloat16)
w = (torch.randn(2560, 320, device=dev) * 0.05).bfloat16()
b = torch.randn(2560, device=dev)
ms = timed(lambda: hip_ops.ff_geglu(x, w, b))
print(f"ff_geglu {rows} x 320 -> 1280: {ms * 1e3:.0f} us {4.0 * rows * 320 * 1280 / ms / 1e9:.0f} TF", flush=True)
for N in (960, 320, 1280):
    wn = (torch.randn(N, 320, device=dev) * 0.05).bfloat16()
    bn = torch.randn(N, device=dev)
    ms = timed(lambda: hip_ops.linear_k320(x, wn, bn))
    print(f"linear_k320 {rows} x 320 -> {N}: {ms * 1e3:.0f} us {2.0 * rows * 320 * N / ms / 1e9:.0f} TF  {rows * (320 + N) * 2 / ms / 1e6:.0f} GB/s", flush=True)
