"""[258048, 320] x [320, 320] + bias (to_out / proj_in / proj_out of the level-0 transformer blocks): the x-stationary kernel of
csrc/ff_geglu.hip (mvi_linear_k320) against the output-stationary kernel of csrc/linear_n320.hip (mvi_linear_n320) and the library."""
import torch
import torch.nn.functional as F
from multiview_inpaint_amd.svd import hip_ops

torch.manual_seed(0)
x = torch.randn(258048, 320, device="cuda", dtype=torch.bfloat16)
w = (torch.randn(320, 320, device="cuda") * 0.05).bfloat16()
b = torch.randn(320, device="cuda")


def timed(fn, n=30):
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


ref = F.linear(x, w, b.bfloat16())
for name, fn in (("linear_k320", lambda: hip_ops.linear_k320(x, w, b)), ("linear_n320", lambda: hip_ops.linear_n320(x, w, b)),
                 ("library", lambda: F.linear(x, w, b.bfloat16()))):
    y = fn()
    print(f"{name}: {timed(fn):.1f} us, maxdiff vs library {(y.float() - ref.float()).abs().max().item():.4f}", flush=True)
