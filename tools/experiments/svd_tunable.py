"""Does PyTorch's TunableOp (GEMM solution search) pay for the SVD step's Linear layers? (GPU box)"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from multiview_inpaint_amd.svd import bench_svd
t0 = time.perf_counter()
r = bench_svd.run_gpu(torch.device("cuda"), steps=2, warmup=int(sys.argv[1]) if len(sys.argv) > 1 else 1, sample_steps=0,
                      weights=sys.argv[2] if len(sys.argv) > 2 else "bf16")
print({k: r[k] for k in ("steps_per_s", "ms_per_step")}, f"total {time.perf_counter() - t0:.1f} s", flush=True)
