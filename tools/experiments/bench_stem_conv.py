"""The two 16-channel layers of the ControlNet hint stem at 28 x 576 x 1024 (bf16): csrc/stem_conv.hip against the library convolution
+ the fused bias + SiLU pass."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from multiview_inpaint_amd.svd import hip_ops, bench_svd
bench_svd.use_shipped_miopen_db()
g = torch.Generator(device="cuda").manual_seed(0)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n
for Cin, Cout, Hh, Ww, st in ((7, 16, 576, 1024, 1), (16, 16, 576, 1024, 1), (16, 32, 576, 1024, 2), (32, 32, 288, 512, 1), (8, 320, 72, 128, 1)):
    x = torch.randn(28, Cin, Hh, Ww, device="cuda", generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * (9 * Cin) ** -0.5).bfloat16()
    b = (torch.randn(Cout, device="cuda", generator=g) * 0.3).bfloat16()
    y = hip_ops.stem_conv3x3_silu(x, w, b, stride=st)
    ref = F.silu(F.conv2d(x, w, b, stride=st, padding=1))
    err = float((y.float() - ref.float()).abs().max() / ref.float().abs().max())
    ms_k = t(lambda: hip_ops.stem_conv3x3_silu(x, w, b, stride=st))
    ms_l = t(lambda: hip_ops.bias_silu(F.conv2d(x, w, None, stride=st, padding=1), b))
    gb = (x.numel() + y.numel()) * 2 / 1e9
    print(f"C_in {Cin:2d} -> {Cout} stride {st} @ 28 x {Hh} x {Ww}: kernel {ms_k * 1e3:7.1f} us ({gb / ms_k * 1e3:5.0f} GB/s)   library conv + bias_silu {ms_l * 1e3:7.1f} us   max |diff| / max {err:.2e}", flush=True)
