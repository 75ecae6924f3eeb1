"""Where does the extra start-up time of the channels-last convolution path go? Times model build, first, second and third step."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from multiview_inpaint_amd.svd import bench_svd
t0 = time.time()
bench_svd.use_shipped_miopen_db()
torch.backends.cudnn.benchmark = os.environ.get("PROBE_BENCHMARK", "1") == "1"
bench_svd.enable_gemm_tuning()
dev = torch.device("cuda")
eng = bench_svd.build(dev, with_control=True, dtype=torch.bfloat16)
x, cond, ind = bench_svd.inputs(dev, 14, 72, 128)
cond = {k: v.bfloat16() for k, v in cond.items()}
sig = torch.full((x.shape[0],), 5.0, device=dev)
kw = dict(num_video_frames=14, image_only_indicator=ind)
torch.cuda.synchronize(); print(f"build {time.time() - t0:.1f} s", flush=True)
for i in range(3):
    t1 = time.time()
    with torch.no_grad():
        eng.denoise(x, sig, cond, **kw)
    torch.cuda.synchronize(); print(f"step {i}: {time.time() - t1:.2f} s", flush=True)
