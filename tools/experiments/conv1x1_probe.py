"""The (3,1,1) temporal convolutions run as 1x1 convolutions with K = 3 C on NCHW tensors (MIOpen) — against the same contraction
as a GEMM on token-major rows (hipBLASLt).  python tools/experiments/conv1x1_probe.py"""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from multiview_inpaint_amd.svd import bench_svd
bench_svd.use_shipped_miopen_db()
bench_svd.enable_gemm_tuning()
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
tot = [0.0, 0.0]
for (n, c, h, w, calls) in [(28, 320, 72, 128, 14), (28, 640, 36, 64, 14), (28, 1280, 18, 32, 14), (28, 1280, 9, 16, 22)]:
    x = torch.randn(n, 3 * c, h, w, device=dev, generator=g).bfloat16()
    wt = (torch.randn(c, 3 * c, device=dev, generator=g) * 0.02).bfloat16()
    tok = x.flatten(2).transpose(1, 2).contiguous()
    res = []
    for fn in (lambda: F.conv2d(x, wt.view(c, 3 * c, 1, 1)), lambda: F.linear(tok, wt)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / 10)
    tot[0] += res[0] * calls; tot[1] += res[1] * calls
    print(f"{3 * c} -> {c} @ {h}x{w}: 1x1 conv (NCHW) {res[0] * 1e3:7.1f} us   linear on tokens {res[1] * 1e3:7.1f} us   x {calls} calls", flush=True)
print(f"per step: {tot[0]:.2f} ms vs {tot[1]:.2f} ms")
