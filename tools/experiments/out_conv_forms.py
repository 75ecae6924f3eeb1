"""The UNet's last convolution (320 -> 4 channels, 3x3, [28, 320, 72, 128] bf16; video_model.py `self.out`) behind a token-major norm: as the
library convolution of a channels-last VIEW of the tokens (with and without bias) against the b c h w convolution behind a layout pass.
GPU box: python tools/experiments/out_conv_forms.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from multiview_inpaint_amd.svd import bench_svd, hip_ops  # noqa: E402

bench_svd.use_shipped_miopen_db()
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
N, C, H, W = 28, 320, 72, 128
t = torch.randn(N, H * W, C, device=dev, generator=g).bfloat16()
w = (torch.randn(4, C, 3, 3, device=dev, generator=g) * 0.02).bfloat16()
b = torch.randn(4, device=dev, generator=g).bfloat16()
wcl = w.contiguous(memory_format=torch.channels_last)


def f_view_bias():
    return F.conv2d(t.view(N, H, W, C).permute(0, 3, 1, 2), wcl, b, 1, 1).contiguous()


def f_view_nobias():
    return (F.conv2d(t.view(N, H, W, C).permute(0, 3, 1, 2), wcl, None, 1, 1) + b.view(1, 4, 1, 1)).contiguous()


def f_planes():
    return F.conv2d(hip_ops.tokens_to_planes_add(t, None, spatial=(H, W)), w, b, 1, 1)


ref = f_planes().float()
for name, fn in [("channels-last view + bias", f_view_bias), ("channels-last view, bias added after", f_view_nobias), ("layout pass + b c h w", f_planes)]:
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            y = fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) / 5)
    print(f"{name}: {sorted(ts)[2] * 1e3:.1f} us, max |diff| vs b c h w {float((y.float() - ref).abs().max()):.3e}", flush=True)
