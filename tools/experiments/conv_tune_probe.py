"""Does MIOpen's exhaustive tuning (MIOPEN_FIND_ENFORCE=SEARCH) find a faster CK instance for the step's 3x3 convolutions than the
default performance config?  MIOPEN_FIND_ENFORCE=3 MIOPEN_USER_DB_PATH=<dir> python tools/experiments/conv_tune_probe.py <idx>"""
import os, sys, time, torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
shapes = [(28, 320, 72, 128, 320), (28, 640, 36, 64, 640), (28, 1280, 18, 32, 1280), (28, 1280, 9, 16, 1280)]
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n, ci, h, w, co = shapes[idx]
x = torch.randn(n, ci, h, w, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
wt = (torch.randn(co, ci, 3, 3, device=dev, generator=g) * 0.02).bfloat16().contiguous(memory_format=torch.channels_last)
t0 = time.time()
y = F.conv2d(x, wt, None, padding=1)
torch.cuda.synchronize()
print(f"first call (find / tuning): {time.time() - t0:.1f} s", flush=True)
for _ in range(3):
    F.conv2d(x, wt, None, padding=1)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    F.conv2d(x, wt, None, padding=1)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print(f"conv {ci}->{co} @ {h}x{w} channels_last: {ms * 1e3:.1f} us ({2.0 * n * h * w * ci * co * 9 / ms * 1e-9:.0f} TF), MIOPEN_FIND_ENFORCE={os.environ.get('MIOPEN_FIND_ENFORCE')}", flush=True)
