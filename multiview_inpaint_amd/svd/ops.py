"""Device ops of the denoise loop. On a GPU tensor these run the hand-written HIP kernels through
the C-ABI library (include/mvi_unet_ops.h) and raise if it is missing — there is no silent
PyTorch substitute on the GPU inference path. On a CPU tensor they are plain PyTorch: that is
BASELINE.json configs[0] ("sgm VideoUNet single denoise step ... fp32 on CPU PyTorch"), the
reference's own CPU-runnable case, not a fallback for the GPU.

The HIP kernels are forward-only. Under autograd (ControlNet training — SURVEY.md §2 row 21, out of
scope) a GPU tensor that requires grad goes through PyTorch-ROCm's differentiable ops instead.

Reference ops: GroupNorm32 + SiLU (sgm/modules/diffusionmodules/util.py:259-276,
openaimodel.py:257-261,292-305), Normalize (sgm/modules/attention.py:125-128),
softmax(QK^T d^-1/2)V (sgm/modules/attention.py:332-336, :427-439).
"""
import torch
import torch.nn.functional as F


def _needs_autograd(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def group_norm(x, num_groups, weight, bias, eps, silu=False):
    """GroupNorm over (C/G, *spatial) with fp32 statistics, optional fused SiLU; output dtype = x.dtype."""
    if x.is_cuda and not _needs_autograd(x, weight, bias):
        from . import hip_ops
        return hip_ops.group_norm_silu(x, num_groups, weight, bias, eps, silu)
    y = F.group_norm(x.float(), num_groups, weight, bias, eps).type(x.dtype)
    return F.silu(y) if silu else y


def attention(q, k, v, heads):
    """q [B,Sq,H*D], k/v [B,Sk,H*D] token-major as the Linear layers produce them -> [B,Sq,H*D]."""
    B, Sq, HD = q.shape
    Sk = k.shape[1]
    if q.is_cuda and not _needs_autograd(q, k, v):
        from . import hip_ops
        return hip_ops.attention(q, k, v, heads)
    D = HD // heads
    qh, kh, vh = (t.reshape(B, -1, heads, D).transpose(1, 2) for t in (q, k, v))
    o = F.scaled_dot_product_attention(qh, kh, vh)
    return o.transpose(1, 2).reshape(B, Sq, HD)
