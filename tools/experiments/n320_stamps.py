"""Phase lengths inside csrc/linear_n320.hip's blocks (diagnostic build: tools/n320_dev/build_stamped.sh -> MVI_HIP_LIB=ab/n320_stamped.so;
MVI_HIP_LIB=ab/n320_stamped.so python tools/experiments/n320_stamps.py). The launcher prints to stderr after every call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multiview_inpaint_amd.svd import hip_ops  # noqa: E402

torch.manual_seed(0)
dev = "cuda"
for (H, W, C, Co) in [(72, 128, 320, 320), (72, 128, 960, 320), (36, 64, 640, 640), (18, 32, 1280, 1280)]:
    tok = torch.randn(28, H * W, C, device=dev, dtype=torch.bfloat16)
    wt = hip_ops.conv3x3_n320_weight((torch.randn(Co, C, 3, 3, device=dev) * 0.02).bfloat16())
    print(f"3x3 {H}x{W} {C}->{Co}", file=sys.stderr, flush=True)
    for _ in range(3):
        hip_ops.conv3x3_n320(tok, wt, None, H, W)
for (rows, K) in [(28 * 9216, 1280), (28 * 9216, 320)]:
    x = torch.randn(rows, K, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(320, K, device=dev) * 0.02).bfloat16()
    print(f"linear {rows} x {K}", file=sys.stderr, flush=True)
    for _ in range(3):
        hip_ops.linear_n320(x, w, None)
torch.cuda.synchronize()
