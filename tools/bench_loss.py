"""Fused photometric loss (fwd + image gradient) at 1920x1080 on the GPU: the HIP path vs the same formula written
with PyTorch-ROCm ops the way gs-simp/utils/loss_utils.py writes it (5 depthwise 11x11 convs + elementwise, autograd).
Usage (GPU box): python tools/bench_loss.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multiview_inpaint_amd import train_ops as T  # noqa: E402

H, W = 1080, 1920
g = torch.Generator(device="cuda").manual_seed(0)
gt = torch.rand(3, H, W, device="cuda", generator=g)
img = (gt + 0.1 * torch.randn(3, H, W, device="cuda", generator=g)).clamp(0, 1)
w1 = torch.tensor([pow(2.718281828459045, -(x - 5) ** 2 / 4.5) for x in range(11)], device="cuda")
w1 = w1 / w1.sum()
win = (w1[:, None] @ w1[None, :]).expand(3, 1, 11, 11).contiguous()


def torch_step():
    x = img.clone().requires_grad_(True)
    conv = lambda t: F.conv2d(t[None], win, padding=5, groups=3)[0]
    mu1, mu2 = conv(x), conv(gt)
    s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(gt * gt) - mu2 * mu2, conv(x * gt) - mu1 * mu2
    smap = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    loss = 0.8 * (x - gt).abs().mean() + 0.2 * (1 - smap.mean())
    loss.backward()
    return x.grad


def hip_step():
    return T.photometric_loss_forward_backward(img, gt, 0.2)[1]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


t_hip, t_torch = timeit(hip_step), timeit(torch_step)
alg = 3 * 3 * H * W * 4
print(f"photometric loss fwd+bwd 1920x1080: HIP {t_hip:.4f} ms ({alg / t_hip / 1e6:.0f} GB/s algorithmic: image + gt in, "
      f"gradient out), PyTorch-ROCm ops {t_torch:.3f} ms  -> {t_torch / t_hip:.1f}x")
