"""Census of the LIBRARY GEMMs (aten::mm / addmm / bmm -> hipBLASLt) left in one full-size SVD denoise step (14 x 576x1024, bf16):
shapes, calls, device time per call and the rate — what csrc/linear_n320.hip's column-group form could take over.
Run on the GPU box:  python tools/gemm_census.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multiview_inpaint_amd.svd import bench_svd, ops  # noqa: E402

dev = torch.device("cuda")
ops.STRICT = True
bench_svd.use_shipped_miopen_db()
bench_svd.enable_gemm_tuning()
eng = bench_svd.build(dev, with_control=True, dtype=torch.bfloat16)
x, cond, ind = bench_svd.inputs(dev, 14, 72, 128)
cond = {k: v.bfloat16() for k, v in cond.items()}
sig = torch.full((x.shape[0],), 5.0, device=dev)
kw = dict(num_video_frames=14, image_only_indicator=ind)
with torch.no_grad():
    for _ in range(2):
        eng.denoise(x, sig, cond, **kw)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    with torch.no_grad():
        eng.denoise(x, sig, cond, **kw)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    if ev.key not in ("aten::mm", "aten::addmm", "aten::bmm") or ev.self_device_time_total <= 0:
        continue
    sh = [s for s in ev.input_shapes if len(s) >= 2]
    if ev.key == "aten::addmm":
        a, b = sh[-2], sh[-1]
    else:
        a, b = sh[0], sh[1]
    M, K, N = a[-2], a[-1], b[-1]
    batch = a[0] if len(a) == 3 else 1
    fl = 2.0 * batch * M * K * N
    us = ev.self_device_time_total / ev.count
    rows.append((ev.self_device_time_total, ev.count, ev.key, batch, M, K, N, us, fl / us / 1e6))
rows.sort(reverse=True)
print(f"library GEMMs in one step: {sum(r[0] for r in rows) / 1e3:.2f} ms (profiler-inflated)")
print(f"{'op':12s} {'batch':>5s} {'M':>7s} {'K':>6s} {'N':>6s} {'calls':>5s} {'us/call':>8s} {'TFLOP/s':>8s} {'ms':>7s}")
for tot, n, op, batch, M, K, N, us, tf in rows:
    print(f"{op:12s} {batch:5d} {M:7d} {K:6d} {N:6d} {n:5d} {us:8.1f} {tf:8.0f} {tot / 1e3:7.2f}")
