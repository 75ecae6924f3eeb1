"""Launch time of csrc/linear_n320.hip against K at [258048, K] x [K, 320] (plain) and for the 3x3 convolution against C_in:
the intercept is the per-launch cost that does not scale with the contraction (1008 blocks = 3.94 rounds on 256 CUs)."""
import torch
from multiview_inpaint_amd.svd import hip_ops

torch.manual_seed(0)


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
    return e0.elapsed_time(e1) / n * 1e3


for K in (128, 192, 320, 640, 1280):
    x = torch.randn(258048, K, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(320, K, device="cuda") * 0.05).bfloat16()
    print(f"plain K {K}: {timed(lambda: hip_ops.linear_n320(x, w, None)):.1f} us", flush=True)
for C in (64, 128, 320):
    tok = torch.randn(28, 72 * 128, C, device="cuda", dtype=torch.bfloat16)
    wt = hip_ops.conv3x3_n320_weight((torch.randn(320, C, 3, 3, device="cuda") * 0.02).bfloat16())
    print(f"conv C_in {C} (K {9 * C}): {timed(lambda: hip_ops.conv3x3_n320(tok, wt, None, 72, 128)):.1f} us", flush=True)
