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
# round 6: the GEGLU form on the PLAIN grid (MVI_N320_PERSIST=0: a block = a tile, which is what the stamps record) — what the persistent
# grid has to hide: the first x / W chunk of a tile (3.5 - 4.7 us beside a 17 - 18 us loop at K = 640). The GEGLU branch leaves before
# the last stamp: its 'epilogue' and 'gap' columns are not meaningful, tile time - loop - first chunk is (6.6 us at K = 640)
os.environ["MVI_N320_PERSIST"] = "0"
for (rows, K, inner) in [(28 * 2304, 640, 2560), (28 * 576, 1280, 5120)]:
    x = torch.randn(rows, K, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(2 * inner, K, device=dev) * K ** -0.5).bfloat16()
    b = torch.randn(2 * inner, device=dev).bfloat16()
    print(f"geglu {rows} x {K} -> 2 x {inner} (plain grid)", file=sys.stderr, flush=True)
    for _ in range(3):
        hip_ops.ff_geglu_n320(x, w, b)
torch.cuda.synchronize()
