"""What a read-only pass reaches on this box: torch reductions over the level-0 activation (165 MB bf16) and over a tensor past the
256 MB Infinity Cache, next to the statistics kernels of the GroupNorm forms (which read the same bytes)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n
for shape in [(28, 9216, 320), (28, 9216, 640), (4, 28, 9216, 320)]:
    x = torch.randn(shape, device="cuda").bfloat16()
    gb = x.numel() * 2 / 1e9
    for name, fn in [("sum", lambda: x.sum()), ("amax", lambda: x.amax()), ("sum(-1)", lambda: x.sum(-1)), ("view int32 sum", lambda: x.view(torch.int32).sum()),
                     ("copy_", lambda: y.copy_(x))]:
        if name == "copy_": y = torch.empty_like(x)
        ms = t(fn)
        print(f"{str(shape):22s} {name:16s} {ms*1e3:8.1f} us  {gb/ms*1e3:7.0f} GB/s read" + (f" (+ same written)" if name == "copy_" else ""), flush=True)
