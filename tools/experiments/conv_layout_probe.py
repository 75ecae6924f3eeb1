"""3x3 convolutions of the SVD step as MIOpen runs them from NCHW tensors (NHWC kernel + transposes around it) and from
channels-last tensors (PYTORCH_MIOPEN_SUGGEST_NHWC=1): is the difference the transposes?  python tools/experiments/conv_layout_probe.py"""
import os
os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
for (n, ci, h, w, co) in [(28, 320, 72, 128, 320), (28, 640, 36, 64, 640), (28, 1280, 18, 32, 1280), (28, 1280, 9, 16, 1280), (28, 640, 72, 128, 320)]:
    x = torch.randn(n, ci, h, w, device=dev, generator=g).bfloat16()
    wt = (torch.randn(co, ci, 3, 3, device=dev, generator=g) * 0.02).bfloat16()
    xcl, wcl = x.contiguous(memory_format=torch.channels_last), wt.contiguous(memory_format=torch.channels_last)
    res = []
    for a, b in ((x, wt), (xcl, wcl)):
        for _ in range(3):
            y = F.conv2d(a, b, None, padding=1)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            y = F.conv2d(a, b, None, padding=1)
        e.record()
        torch.cuda.synchronize()
        res.append((s.elapsed_time(e) / 10, y))
    fl = 2.0 * n * h * w * ci * co * 9
    d = float((res[0][1].float() - res[1][1].float()).abs().max())
    print(f"conv {ci}->{co} @ {h}x{w}: NCHW {res[0][0] * 1e3:7.1f} us ({fl / res[0][0] * 1e-9:5.0f} TF)   channels_last {res[1][0] * 1e3:7.1f} us ({fl / res[1][0] * 1e-9:5.0f} TF)"
          f"   out is channels_last: {res[1][1].is_contiguous(memory_format=torch.channels_last)}  max diff {d:.3g}", flush=True)
