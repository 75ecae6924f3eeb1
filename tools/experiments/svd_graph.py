"""Does one SVD denoise step capture into a HIP graph, does the replay match eager, and what does it save?
Run on the GPU box: timeout -k 10 600 python tools/experiments/svd_graph.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from multiview_inpaint_amd.svd import bench_svd  # noqa: E402
from multiview_inpaint_amd.svd.schedule import EDMDiscretization  # noqa: E402

dev = torch.device("cuda")
torch.backends.cudnn.benchmark = True
bench_svd.use_shipped_miopen_db()
bench_svd.enable_gemm_tuning()
eng = bench_svd.build(dev, dtype=torch.bfloat16)
x, cond, ind = bench_svd.inputs(dev)
cond = {k: v.bfloat16() for k, v in cond.items()}
sig = EDMDiscretization(sigma_max=700.0)(25, device=dev)
kw = dict(num_video_frames=14, image_only_indicator=ind)
xs, ss = x.clone(), sig[0].expand(x.shape[0]).clone()


def eager(i):
    with torch.no_grad():
        return eng.denoise(x, sig[i % 25].expand(x.shape[0]), cond, **kw)


for i in range(2):
    eager(i)
torch.cuda.synchronize()
print("eager warm", flush=True)
t0 = time.perf_counter()
for i in range(3):
    ref = eager(i)
torch.cuda.synchronize()
print(f"eager {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms/step", flush=True)

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side), torch.no_grad():
    for _ in range(2):
        eng.denoise(xs, ss, cond, **kw)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("side-stream warm", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side), torch.no_grad():
    out = eng.denoise(xs, ss, cond, **kw)
torch.cuda.synchronize()
print("captured", flush=True)
for i in range(2):
    ss.copy_(sig[i].expand(x.shape[0]))
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3):
    ss.copy_(sig[i % 25].expand(x.shape[0]))
    g.replay()
torch.cuda.synchronize()
print(f"graph {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms/step", flush=True)
d = (out.float() - ref.float()).abs().max().item()
print(f"max |graph - eager| at sigma[2]: {d:.3e} (ref max {ref.float().abs().max().item():.3e}); bit-equal: {torch.equal(out, ref)}")
