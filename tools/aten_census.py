"""Census of the torch (ATen) elementwise / copy kernels left in one full-size SVD denoise step (14 x 576x1024, bf16):
op, input shapes, calls, device time, and the Python frame that issued it — what is still outside the HIP ops.
Run on the GPU box:  python tools/aten_census.py"""
import collections
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    with torch.no_grad():
        eng.denoise(x, sig, cond, **kw)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
    if not ev.key.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    if any(s in ev.key for s in ("convolution", "miopen", "mm", "linear")):
        continue
    frame = next((s for s in (ev.stack or []) if "multiview_inpaint_amd" in s), "?")
    rows.append((ev.self_device_time_total, ev.count, ev.key, str(ev.input_shapes)[:90], frame.split("multiview_inpaint_amd/")[-1][:70]))
tot = sum(r[0] for r in rows)
print(f"{len(rows)} distinct ATen calls with device time of their own, {tot / 1e3:.2f} ms per step (profiler-inflated)")
for us, n, name, shapes, frame in sorted(rows, reverse=True)[:45]:
    print(f"{us / 1e3:7.3f} ms {n:4d}x {name:24s} {shapes:90s} {frame}")
