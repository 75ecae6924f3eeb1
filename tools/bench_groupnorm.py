"""GroupNorm(+SiLU) micro-benchmark at the shapes of the 14 x 576x1024 step, bf16:
MVI_GN_RESIDENT=0 python tools/bench_groupnorm.py   (two-launch form)   vs   python tools/bench_groupnorm.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multiview_inpaint_amd.svd import hip_ops
import torch.nn.functional as F
only = os.environ.get("GN_SHAPE")            # e.g. GN_SHAPE=28,320,72,128: that shape only (per-kernel times under rocprofv3 --stats)
only = tuple(int(v) for v in only.split(",")) if only else None
shapes = [(28, 320, 72, 128), (28, 640, 72, 128), (28, 640, 36, 64), (28, 1280, 36, 64), (28, 1920, 36, 64), (28, 1280, 18, 32),
          (28, 2560, 18, 32), (28, 1280, 9, 16)]
g = torch.Generator(device="cuda").manual_seed(0)
for shp in shapes:
    if only and shp != only:
        continue
    x = (torch.randn(shp, device="cuda", generator=g) * 1.5 + 0.3).bfloat16()
    w, b = torch.randn(shp[1], device="cuda", generator=g), torch.randn(shp[1], device="cuda", generator=g)
    cb = torch.randn(shp[0], shp[1], device="cuda", generator=g)
    y = hip_ops.group_norm_silu(x, 32, w, b, 1e-5, True, chan_bias=cb)
    ref = F.silu(F.group_norm(x.float() + cb[:, :, None, None], 32, w, b, 1e-5))
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    for _ in range(3):
        hip_ops.group_norm_silu(x, 32, w, b, 1e-5, True, chan_bias=cb)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        hip_ops.group_norm_silu(x, 32, w, b, 1e-5, True, chan_bias=cb)
    e.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(e) / 20
    gb = 2 * x.numel() * 2 / 1e9
    print(f"{str(shp):22s} {ms * 1e3:8.1f} us  {gb / ms * 1e3:7.0f} GB/s algorithmic ({gb / ms * 1e3 / 8000:.3f} of 8 TB/s)  rel err {err:.2e}", flush=True)

# token-major forms used inside the ResBlocks when their convolutions run channels-last (planes -> tokens, tokens -> tokens)
print("planes -> tokens (mvi_groupnorm_silu_tokens) | tokens -> tokens (mvi_groupnorm_silu_tok2tok), fused bias + SiLU")
for shp in [(28, 320, 72, 128), (28, 640, 72, 128), (28, 640, 36, 64), (28, 1280, 18, 32), (28, 1280, 9, 16)]:
    if only and shp != only:
        continue
    x = (torch.randn(shp, device="cuda", generator=g) * 1.5 + 0.3).bfloat16()
    t = x.flatten(2).transpose(1, 2).contiguous()
    w, b = torch.randn(shp[1], device="cuda", generator=g), torch.randn(shp[1], device="cuda", generator=g)
    cb = torch.randn(shp[0], shp[1], device="cuda", generator=g)
    res = []
    for fn in (lambda: hip_ops.group_norm_silu_tokens(x, 32, w, b, 1e-5, True, chan_bias=cb), lambda: hip_ops.group_norm_silu_tok2tok(t, 32, w, b, 1e-5, True, chan_bias=cb)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(e) / 20)
    gb = 2 * x.numel() * 2 / 1e9
    print(f"{str(shp):22s} {res[0] * 1e3:8.1f} us {gb / res[0] * 1e3:6.0f} GB/s | {res[1] * 1e3:8.1f} us {gb / res[1] * 1e3:6.0f} GB/s", flush=True)
