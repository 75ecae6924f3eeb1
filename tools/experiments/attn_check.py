"""Quick numerical check of the attention kernel build in place (GPU box): bf16 vs fp64 reference."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from multiview_inpaint_amd.svd import hip_ops
import torch.nn.functional as F
g = torch.Generator().manual_seed(0)
for (B, H, S, D) in [(2, 5, 576, 64), (1, 2, 200, 64), (1, 1, 9216, 64)]:
    q, k, v = (torch.randn(B, S, H * D, generator=g).bfloat16() for _ in range(3))
    ref = F.scaled_dot_product_attention(*(t.double().reshape(B, S, H, D).transpose(1, 2) for t in (q, k, v))).transpose(1, 2).reshape(B, S, H * D)
    out = hip_ops.attention(q.cuda(), k.cuda(), v.cuda(), H).double().cpu()
    print(f"B{B} H{H} S{S}: rel err {float((out - ref).abs().max() / ref.abs().max()):.3e}", flush=True)
