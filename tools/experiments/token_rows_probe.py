"""What the batched cross-attention rows (svd/transformer.py prepare_single_token_rows) cost per network call at full size: the one to_v GEMM
over the concatenated weights and the batched to_out GEMMs per width, against the 2 x 46 per-layer GEMMs they replace. GPU box:
python tools/experiments/token_rows_probe.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from multiview_inpaint_amd.svd import bench_svd  # noqa: E402

bench_svd.enable_gemm_tuning()
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.03).bfloat16()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3


# UNet of configs[3]: per level (320, 640, 1280) the spatial + temporal transformer blocks: 7 / 7 / 9 layers of each kind (in + out + middle)
layers = {320: 7, 640: 7, 1280: 9}
for n in (28, 2):
    ctx = rn(n, 1024)
    Wv = rn(sum(c * l for c, l in layers.items()), 1024)
    t_v = timed(lambda: F.linear(ctx, Wv))
    V = F.linear(ctx, Wv)
    tot = t_v
    line = [f"to_v of all layers [{n}, 1024] x [{Wv.shape[0]}, 1024]: {t_v:.1f} us"]
    off = 0
    for c, l in layers.items():
        Wo, bo = rn(l, c, c), rn(l, c)
        Vg = V[:, off:off + c * l].reshape(n, l, c).transpose(0, 1)
        t = timed(lambda: torch.baddbmm(bo[:, None, :], Vg, Wo.transpose(1, 2)))
        tot += t
        off += c * l
        line.append(f"to_out x {l} at {c}: {t:.1f} us")
    per = 0.0
    for c, l in layers.items():
        wv, wo, b = rn(c, 1024), rn(c, c), rn(c)
        x3 = ctx.view(n, 1, 1024)
        per += l * timed(lambda: F.linear(F.linear(x3, wv), wo, b))
    print(f"n = {n}: " + "; ".join(line) + f"; batched total {tot:.1f} us against {per:.1f} us for the per-layer pairs", flush=True)
