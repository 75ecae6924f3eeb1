"""FusedAdam.step over the parameter set of a 1.5 M-Gaussian model (59 floats per Gaussian in six groups, gaussian_model.py:154-163):
microseconds per step and the HBM rate of its 28 bytes per parameter. MVI_HIP_LIB selects another build of the library (A/B)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_inpaint_amd import train_ops as T

N = int(os.environ.get("N", 1_500_000))
dev = "cuda"
shapes = dict(xyz=(N, 3), f_dc=(N, 1, 3), f_rest=(N, 15, 3), opacity=(N, 1), scaling=(N, 3), rotation=(N, 4))
g = torch.Generator(dev).manual_seed(0)
prm = {k: torch.nn.Parameter(torch.randn(*s, device=dev, generator=g)) for k, s in shapes.items()}
opt = T.FusedAdam([{"params": [p], "lr": 1e-3, "name": k} for k, p in prm.items()], lr=0.0, eps=1e-15)
for p in prm.values():
    p.grad = torch.randn_like(p)
for _ in range(5):
    opt.step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 50
a.record()
for _ in range(K):
    opt.step()
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) / K * 1e3
n = sum(p.numel() for p in prm.values())
print(json.dumps(dict(lib=os.environ.get("MVI_HIP_LIB", "shipped"), us_per_step=round(us, 1), params=n, TBps=round(n * 28 / us / 1e6, 3),
                      checksum=float(sum(p.detach().double().sum() for p in prm.values())))))
