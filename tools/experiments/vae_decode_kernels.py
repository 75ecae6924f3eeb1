"""Which kernels make up the first-stage decode (14 frames, 72x128 latents -> 576x1024) — torch.profiler's device-time table of one decode
after a warm-up: the split convolutions and norms of this library beside the PyTorch / vendor kernels of the walk (svd/vae_split.py).
    python tools/experiments/vae_decode_kernels.py [--dtype fp32|bf16|f16]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import svd_helpers as H  # noqa: E402
from bench_vae import FULL  # noqa: E402
from multiview_inpaint_amd.svd import vae  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp32")
a = ap.parse_args()
dt = {"fp32": None, "bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
eng = vae.AutoencodingEngine(encoder_config=vae.Encoder(**FULL), decoder_config=vae.VideoDecoder(**FULL, video_kernel_size=[3, 1, 1])).eval()
eng.decoder.load_state_dict(H.seeded_state_dict(eng.decoder, 42))
eng = eng.to("cuda:0")
z = (torch.randn(14, 4, 72, 128, generator=torch.Generator().manual_seed(0)) * 0.18215).to("cuda:0")
with torch.no_grad():
    vae.decode_first_stage(eng, z, dtype=dt)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
        vae.decode_first_stage(eng, z, dtype=dt)
        torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    t = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
    if t and getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower():
        rows.append((t / 1e3, e.count, e.key[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"decode ({a.dtype}): {tot:.1f} ms of device time in {sum(r[1] for r in rows)} kernels")
for ms, n, k in rows[:25]:
    print(f"{ms:9.2f} ms {n:5d}  {k}")
