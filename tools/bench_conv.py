"""Vendor-library 3x3 convolution rates at the SVD UNet shapes (bf16), under the layout / autotune switches
PyTorch exposes. Usage (GPU box): python tools/bench_conv.py"""

import torch
import torch.nn.functional as F

SHAPES = [(28, 320, 320, 72, 128), (28, 640, 320, 72, 128), (28, 640, 640, 36, 64), (28, 1280, 1280, 18, 32),
          (28, 1280, 1280, 9, 16), (28, 2560, 1280, 18, 32)]


def run(tag, channels_last, benchmark):
    torch.backends.cudnn.benchmark = benchmark
    for N, Ci, Co, H, W in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn(N, Ci, H, W, device="cuda", generator=g).bfloat16()
        w = (torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) * 0.02).bfloat16()
        if channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
            w = w.contiguous(memory_format=torch.channels_last)
        for _ in range(3):
            y = F.conv2d(x, w, None, padding=1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            y = F.conv2d(x, w, None, padding=1)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        fl = 2.0 * N * H * W * Ci * Co * 9
        print(f"{tag:28s} N{N} {Ci:4d}->{Co:4d} {H}x{W}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TFLOP/s  out_cl={y.is_contiguous(memory_format=torch.channels_last)}", flush=True)


for cl in (False, True):
    for bm in (False, True):
        run(f"channels_last={cl} bench={bm}", cl, bm)
