"""cProfile of bench_svd.run_gpu (GPU box): where do the timed steps spend host time?"""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from multiview_inpaint_amd.svd import bench_svd
pr = cProfile.Profile(); pr.enable()
r = bench_svd.run_gpu(torch.device("cuda"), steps=2, warmup=int(sys.argv[1]) if len(sys.argv) > 1 else 1, sample_steps=0)
pr.disable()
print({k: r[k] for k in ("steps_per_s", "ms_per_step")}, flush=True)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
