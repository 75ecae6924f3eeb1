"""A few launches of the implicit-GEMM convolution at one level-0 shape (28 x 72x128, 640 -> 320) for counter passes."""
import torch
from multiview_inpaint_amd.svd import hip_ops

torch.manual_seed(0)
N, H, W, C, Co = 28, 72, 128, 640, 320
tok = torch.randn(N, H * W, C, device="cuda", dtype=torch.bfloat16)
wt = hip_ops.conv3x3_n320_weight((torch.randn(Co, C, 3, 3, device="cuda") * 0.02).bfloat16())
for _ in range(6):
    hip_ops.conv3x3_n320(tok, wt, None, H, W)
torch.cuda.synchronize()
