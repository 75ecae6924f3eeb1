"""Temporal GroupNorm(+SiLU) with the stacked (t-1 | t | t+1) output at the shapes of the 14 x 576x1024 step: time per call and
algorithmic rate (x read once + 3 x written)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from multiview_inpaint_amd.svd import hip_ops
g = torch.Generator(device="cuda").manual_seed(0)
for shp in [(28, 320, 72, 128), (28, 640, 36, 64), (28, 1280, 18, 32), (28, 1280, 9, 16)]:
    x = (torch.randn(shp, device="cuda", generator=g) * 1.5 + 0.3).bfloat16()
    w, b = torch.randn(shp[1], device="cuda", generator=g), torch.randn(shp[1], device="cuda", generator=g)
    cb = torch.randn(shp[0], shp[1], device="cuda", generator=g)
    for stack3 in (True, False):
        fn = lambda: hip_ops.group_norm_silu_frames(x, 14, 32, w, b, 1e-5, True, chan_bias=cb, stack3=stack3)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(e) / 20
        gb = (1 + (3 if stack3 else 1)) * x.numel() * 2 / 1e9
        print(f"{str(shp):22s} stack3 {int(stack3)} {ms * 1e3:8.1f} us  {gb / ms * 1e3:7.0f} GB/s algorithmic", flush=True)
