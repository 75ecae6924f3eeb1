"""Does MIOpen run channels_last bf16 convolutions without its NCHW<->NHWC transposes on this stack? (GPU box)"""
import os, sys, time, torch
from torch.profiler import ProfilerActivity, profile
torch.backends.cudnn.benchmark = True
dev = "cuda"
def run(cl, shape=(28, 320, 72, 128), co=320, k=3):
    x = torch.randn(*shape, device=dev, dtype=torch.bfloat16)
    conv = torch.nn.Conv2d(shape[1], co, k, padding=k // 2).to(dev).to(torch.bfloat16)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
        conv = conv.to(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3):
            y = conv(x)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(5):
                y = conv(x)
            torch.cuda.synchronize()
    rows = sorted(((e.self_device_time_total / 5e3, e.key[:90]) for e in prof.key_averages() if e.self_device_time_total > 0), reverse=True)
    print(f"channels_last={cl} shape={shape} co={co} k={k} out_cl={y.is_contiguous(memory_format=torch.channels_last)} total {sum(r[0] for r in rows):.3f} ms", flush=True)
    for ms, name in rows[:5]:
        print(f"    {ms:7.3f} ms  {name}", flush=True)
print("PYTORCH_MIOPEN_SUGGEST_NHWC =", os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC"))
for cl in (False, True):
    run(cl)
    run(cl, (28, 1280, 18, 32), 1280, 3)
    run(cl, (28, 960, 72, 128), 320, 1)
