"""Which aten ops (with input shapes) own the non-GEMM kernel time of one steady-state SVD denoise step.
Usage (GPU box): python tools/diag_svd_ops.py [n_rows]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multiview_inpaint_amd.svd import bench_svd  # noqa: E402
from multiview_inpaint_amd.svd.schedule import EDMDiscretization  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda")
eng = bench_svd.build(dev, dtype=torch.bfloat16)
x, cond, ind = bench_svd.inputs(dev)
cond = {k: v.bfloat16() for k, v in cond.items()}
sig = EDMDiscretization(sigma_max=700.0)(25, device=dev)
kw = dict(num_video_frames=14, image_only_indicator=ind)


def step(i):
    with torch.no_grad():
        return eng.denoise(x, sig[i].expand(x.shape[0]), cond, **kw)


for i in range(2):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(2)
    torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = e.self_cuda_time_total
    if t > 0 and (e.key.startswith("aten::") or len(sys.argv) > 2):
        rows.append((t / 1e3, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total self device time {tot:.1f} ms")
for ms, cnt, key, shp in rows[:n]:
    print(f"{ms:8.2f} ms {cnt:4d}  {key:34s} {shp}")
