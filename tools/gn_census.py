"""Census of the GroupNorm(+SiLU) calls of one full-size SVD denoise step (14 x 576x1024, bf16): shape, statistics span,
flags, calls and time per distinct call — which calls hold the bytes, which kernel form each takes.
Run on the GPU box:  python tools/gn_census.py"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multiview_inpaint_amd.svd import bench_svd, hip_ops, ops  # noqa: E402

dev = torch.device("cuda")
ops.STRICT = True
torch.backends.cudnn.benchmark = True
bench_svd.use_shipped_miopen_db()
bench_svd.enable_gemm_tuning()
eng = bench_svd.build(dev, with_control=True, dtype=torch.bfloat16)
x, cond, ind = bench_svd.inputs(dev, 14, 72, 128)
cond = {k: v.bfloat16() for k, v in cond.items()}
sig = torch.full((x.shape[0],), 5.0, device=dev)
kw = dict(num_video_frames=14, image_only_indicator=ind)
with torch.no_grad():
    eng.denoise(x, sig, cond, **kw)
torch.cuda.synchronize()
rec = collections.OrderedDict()
orig = hip_ops._gn


def patched(x, T, num_groups, weight, bias, eps, silu, chan_bias, stack3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    y = orig(x, T, num_groups, weight, bias, eps, silu, chan_bias, stack3)
    b.record()
    b.synchronize()
    N, C = x.shape[0], x.shape[1]
    S = x.numel() // (N * C)
    key = (tuple(x.shape), int(T), bool(stack3), chan_bias is not None, bool(silu))
    group_kb = (C // num_groups) * S * T * x.element_size() / 1024
    byt = (2 + 2 * bool(stack3)) * x.numel() * x.element_size()
    r = rec.setdefault(key, [0, 0.0, group_kb, byt])
    r[0] += 1
    r[1] += a.elapsed_time(b)
    return y


hip_ops._gn = patched
with torch.no_grad():
    eng.denoise(x, sig, cond, **kw)
torch.cuda.synchronize()
tot = sum(r[1] for r in rec.values())
print(f"{len(rec)} distinct GroupNorm calls, {sum(r[0] for r in rec.values())} calls, {tot:.2f} ms per step (event-bracketed, one call at a time)")
print(f"{'shape':28s} {'T':>3s} stack3 bias silu {'group KB':>9s} {'calls':>5s} {'ms':>7s} {'us/call':>8s} {'GB/s alg':>9s}")
for (shape, T, s3, cb, silu), (n, ms, gkb, byt) in sorted(rec.items(), key=lambda kv: -kv[1][1]):
    print(f"{str(shape):28s} {T:3d} {int(s3):6d} {int(cb):4d} {int(silu):4d} {gkb:9.0f} {n:5d} {ms:7.3f} {ms / n * 1e3:8.1f} {byt * n / ms * 1e-6:9.0f}")
