import torch, time, sys
import torch.nn.functional as F
from multiview_inpaint_amd.svd import hip_ops
torch.manual_seed(0)
dev = "cuda"
for (N, H, W, C) in [(28, 72, 128, 320), (28, 72, 128, 640), (28, 72, 128, 960)]:
    tok = torch.randn(N, H * W, C, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(320, C, 3, 3, device=dev) * 0.02).bfloat16()
    wt = hip_ops.conv3x3_n320_weight(w)
    wcl = w.contiguous(memory_format=torch.channels_last)
    x = tok.view(N, H, W, C).permute(0, 3, 1, 2)
    def lib():
        return F.conv2d(x, wcl, None, 1, 1)
    def mine():
        return hip_ops.conv3x3_n320(tok, wt, None, H, W)
    a = lib().permute(0, 2, 3, 1).reshape(N, H * W, 320); b = mine()
    print("maxdiff", (a.float() - b.float()).abs().max().item(), "ref max", a.float().abs().max().item())
    for name, fn in (("miopen", lib), ("conv3x3_n320", mine)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        fl = 2.0 * N * H * W * 9 * C * 320
        print(f"C_in {C} {name}: {ms*1e3:.0f} us  {fl/ms/1e9:.0f} TFLOP/s", flush=True)
