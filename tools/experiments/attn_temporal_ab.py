"""Temporal attention at the four levels of the 14-frame 576x1024 step: GB/s of algorithmic bytes (q, k, v read, out written once).
MVI_ATTN_TEMPORAL_MFMA=0 selects the fp32-math kernel of csrc/attn_rowtile.hip for the A/B."""
import torch
from multiview_inpaint_amd.svd import hip_ops

torch.manual_seed(0)
T, bo = 14, 2
for (S, H) in [(72 * 128, 5), (36 * 64, 10), (18 * 32, 20), (9 * 16, 20)]:
    qkv = torch.randn(bo * T, S, 3 * H * 64, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        hip_ops.attention_temporal_packed(qkv, H, T)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip_ops.attention_temporal_packed(qkv, H, T)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nbytes = 4.0 * bo * T * S * H * 64 * 2
    print(f"S {S} H {H}: {ms * 1e3:.1f} us, {nbytes / ms / 1e6:.0f} GB/s ({nbytes / ms / 1e6 / 8000:.2f} of 8 TB/s)", flush=True)
