"""Where does the host spend its time in one steady-state SVD denoise step? (GPU box) cProfile of step 2."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from multiview_inpaint_amd.svd import bench_svd
from multiview_inpaint_amd.svd.schedule import EDMDiscretization
torch.backends.cudnn.benchmark = "--nobench" not in sys.argv
dev = torch.device("cuda")
eng = bench_svd.build(dev, dtype=torch.bfloat16)
x, cond, ind = bench_svd.inputs(dev)
cond = {k: v.bfloat16() for k, v in cond.items()}
sig = EDMDiscretization(sigma_max=700.0)(25, device=dev)
kw = dict(num_video_frames=14, image_only_indicator=ind)
def step(i):
    with torch.no_grad():
        return eng.denoise(x, sig[i].expand(x.shape[0]), cond, **kw)
for i in range(3):
    t0 = time.perf_counter(); step(i); torch.cuda.synchronize(); print(f"step {i}: {time.perf_counter() - t0:.3f} s", flush=True)
pr = cProfile.Profile(); pr.enable(); step(3); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
