"""Times csrc/linear_n320.hip's forms at the SVD shapes of one denoise step (bf16, hipEvents on the launch stream, no library
comparison, nothing checked: the parity tests do that). For A/B builds: MVI_HIP_LIB=ab/<name>.so python tools/bench_n320.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_inpaint_amd.svd import hip_ops  # noqa: E402

torch.manual_seed(0)
dev = "cuda"
BT, T = 28, 14
if os.environ.get("MVI_BENCH_ZEROS") == "1":        # all-zero operands: what the same instruction stream does when the chip is not held back by power
    _randn = torch.randn
    torch.randn = lambda *a, **k: _randn(*a, **k) * 0


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


total = 0.0
for (H, W, C, Co, calls) in [(72, 128, 320, 320, 18), (72, 128, 640, 320, 4), (72, 128, 960, 320, 2), (36, 64, 640, 640, 14),
                             (36, 64, 1280, 640, 3), (18, 32, 1280, 1280, 14)]:
    tok = torch.randn(BT, H * W, C, device=dev, dtype=torch.bfloat16)
    wt = hip_ops.conv3x3_n320_weight((torch.randn(Co, C, 3, 3, device=dev) * 0.02).bfloat16())
    ms = timed(lambda: hip_ops.conv3x3_n320(tok, wt, None, H, W))
    fl = 2.0 * BT * H * W * 9 * C * Co
    total += ms * calls
    print(f"3x3 {H}x{W} {C}->{Co}: {ms * 1e3:.0f} us {fl / ms / 1e9:.0f} TF", flush=True)
for (H, W, C, calls) in [(72, 128, 320, 24), (36, 64, 640, 20), (18, 32, 1280, 20)]:
    tok = torch.randn(BT, H * W, C, device=dev, dtype=torch.bfloat16)
    wt = hip_ops.conv3t_n320_weight((torch.randn(C, C, 3, 1, 1, device=dev) * 0.03).bfloat16())
    ms = timed(lambda: hip_ops.conv3t_n320(tok, wt, None, T))
    fl = 2.0 * BT * H * W * 3 * C * C
    total += ms * calls
    print(f"3t {H}x{W} {C}: {ms * 1e3:.0f} us {fl / ms / 1e9:.0f} TF", flush=True)
for (rows, K, calls) in [(BT * 9216, 1280, 21), (BT * 9216, 320, 28)]:
    x = torch.randn(rows, K, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(320, K, device=dev) * 0.02).bfloat16()
    ms = timed(lambda: hip_ops.linear_n320(x, w, None))
    total += ms * calls
    print(f"linear {rows} x {K} -> 320: {ms * 1e3:.0f} us {2.0 * rows * K * 320 / ms / 1e9:.0f} TF", flush=True)
for (rows, K, N, calls) in [(BT * 2304, 2560, 640, 21), (BT * 576, 5120, 1280, 21), (BT * 2304, 640, 640, 28), (BT * 576, 1280, 1280, 28),
                            (BT * 2304, 640, 1920, 14), (BT * 576, 1280, 3840, 14), (BT * 144, 5120, 1280, 6), (BT * 144, 1280, 1280, 8)]:
    x = torch.randn(rows, K, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    b = torch.randn(N, device=dev)
    if hip_ops.linear_n320_supported(K, N, x.dtype):
        ms = timed(lambda: hip_ops.linear_n320(x, w, b))
        ms_l = timed(lambda: torch.nn.functional.linear(x, w, b.bfloat16()))
        print(f"linear {rows} x {K} -> {N}: {ms * 1e3:.0f} us {2.0 * rows * K * N / ms / 1e9:.0f} TF   | library {ms_l * 1e3:.0f} us "
              f"{2.0 * rows * K * N / ms_l / 1e9:.0f} TF  ({calls} calls per step: {(ms_l - ms) * calls:+.2f} ms)", flush=True)
for (rows, K, calls) in [(BT * 2304, 640, 21)]:
    x = torch.randn(rows, K, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(8 * K, K, device=dev) * 0.02).bfloat16()
    ms = timed(lambda: hip_ops.ff_geglu_n320(x, w, None))
    total += ms * calls
    print(f"ff_geglu_n320 {rows} x {K} -> {4 * K}: {ms * 1e3:.0f} us {4.0 * rows * K * 4 * K / ms / 1e9:.0f} TF", flush=True)
print(f"weighted by rough calls per step: {total:.2f} ms", flush=True)
