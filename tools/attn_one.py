import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multiview_inpaint_amd.svd import hip_ops
B, H, S, D = 28, 5, 9216, 64
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn(B, S, H * D, device="cuda", generator=g).to(torch.bfloat16) for _ in range(3))
for _ in range(3):
    hip_ops.attention(q, k, v, H)
torch.cuda.synchronize()
