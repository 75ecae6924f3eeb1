"""A/B of csrc/linear_n320.hip's implicit-GEMM convolutions against the library (MIOpen on channels-last views) at the SVD shapes."""
import os

import torch
import torch.nn.functional as F
from multiview_inpaint_amd.svd import hip_ops

torch.manual_seed(0)
dev = "cuda"
LIB = os.environ.get("AB_LIBRARY", "1") != "0"          # AB_LIBRARY=0: time only the hand-written kernel


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (N, H, W, C, Co) in [(28, 72, 128, 320, 320), (28, 72, 128, 640, 320), (28, 72, 128, 960, 320), (28, 72, 128, 640, 640),
                         (28, 36, 64, 640, 640), (28, 36, 64, 1280, 640), (28, 36, 64, 1280, 1280), (28, 18, 32, 1280, 1280),
                         (28, 18, 32, 2560, 1280), (28, 9, 16, 1280, 1280), (28, 9, 16, 2560, 1280)]:
    tok = torch.randn(N, H * W, C, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(Co, C, 3, 3, device=dev) * 0.02).bfloat16()
    wt = hip_ops.conv3x3_n320_weight(w)
    wcl = w.contiguous(memory_format=torch.channels_last)
    x = tok.view(N, H, W, C).permute(0, 3, 1, 2)
    lib = lambda: F.conv2d(x, wcl, None, 1, 1)
    mine = lambda: hip_ops.conv3x3_n320(tok, wt, None, H, W)
    a = lib().permute(0, 2, 3, 1).reshape(N, H * W, Co)
    b = mine()
    fl = 2.0 * N * H * W * 9 * C * Co
    ms_l, ms_m = (timed(lib) if LIB else float("nan")), timed(mine)
    print(f"3x3 {H}x{W} {C}->{Co}: maxdiff {(a.float() - b.float()).abs().max().item():.3f} of {a.float().abs().max().item():.1f} | "
          f"library {ms_l * 1e3:.0f} us {fl / ms_l / 1e9:.0f} TF | n320 {ms_m * 1e3:.0f} us {fl / ms_m / 1e9:.0f} TF", flush=True)

T = 14
for (H, W, C) in [(72, 128, 320), (36, 64, 640), (18, 32, 1280), (9, 16, 1280)]:
    BT, S = 28, H * W
    tok = torch.randn(BT, S, C, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(C, C, 3, 1, 1, device=dev) * 0.03).bfloat16()
    wt = hip_ops.conv3t_n320_weight(w)
    x5 = tok.view(BT // T, T, H, W, C).permute(0, 4, 1, 2, 3)               # b c t h w view
    ref = F.conv3d(x5, w, None, 1, (1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(BT, S, C)
    x3 = torch.cat([F.pad(x5, (0, 0, 0, 0, 1, 0))[:, :, :T], x5, F.pad(x5, (0, 0, 0, 0, 0, 1))[:, :, 1:]], 1)
    x3 = x3.permute(0, 2, 1, 3, 4).reshape(BT, 3 * C, H, W).contiguous()     # the stacked NCHW input of temporal_conv3_stacked
    w1 = wt.reshape(C, 3 * C, 1, 1)
    lib = lambda: F.conv2d(x3, w1)
    mine = lambda: hip_ops.conv3t_n320(tok, wt, None, T)
    b = mine()
    fl = 2.0 * BT * S * 3 * C * C
    ms_l, ms_m = (timed(lib) if LIB else float("nan")), timed(mine)
    print(f"3t {H}x{W} {C}: maxdiff {(ref.float() - b.float()).abs().max().item():.3f} of {ref.float().abs().max().item():.1f} | "
          f"library 1x1 on stacked NCHW {ms_l * 1e3:.0f} us {fl / ms_l / 1e9:.0f} TF | n320 {ms_m * 1e3:.0f} us {fl / ms_m / 1e9:.0f} TF", flush=True)
