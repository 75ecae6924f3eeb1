"""Token-major GroupNorm at the four levels of the 14-frame 576x1024 step, per-sample and temporal (frames = 14) statistics; run under
rocprofv3 --kernel-trace --stats for the per-kernel split (statistics / merge / apply)."""
import torch
from multiview_inpaint_amd.svd import hip_ops

torch.manual_seed(0)
for (S, C) in [(72 * 128, 320), (36 * 64, 640), (18 * 32, 1280), (9 * 16, 1280)]:
    x = torch.randn(28, S, C, device="cuda", dtype=torch.bfloat16)
    w, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    cb = torch.randn(28, C, device="cuda")
    for frames in (1, 14):
        fn = lambda: hip_ops.group_norm_silu_tok2tok(x, 32, w, b, 1e-5, True, chan_bias=cb, frames=frames)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        nbytes = 2.0 * x.numel() * 2
        print(f"S {S} C {C} frames {frames}: {ms * 1e3:.1f} us, {nbytes / ms / 1e6:.0f} GB/s of 2 x numel ({nbytes / ms / 1e6 / 8000:.2f} of 8 TB/s)", flush=True)
