"""GPU implementations of multiview_inpaint_amd.svd.ops through the C-ABI HIP library
(include/mvi_unet_ops.h). Importing this module without libmvi_hip.so raises."""
import ctypes as C

import torch

from .. import _lib

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
_ws = {}


def _check(rc, what):
    if rc != 0:
        msg = _lib.lib().mvi_unet_last_error().decode(errors="replace")
        raise (Exception if rc == -1 else RuntimeError)(f"{what} failed ({rc}): {msg}")


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _workspace(dev, nbytes):
    """Per-device grow-only scratch (stream-ordered reuse: every op runs on the current stream)."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _ws[key] = w
    return w


def group_norm_silu(x, num_groups, weight, bias, eps, silu):
    L = _lib.lib()
    if x.dtype not in _DT:
        raise TypeError(f"group_norm: unsupported dtype {x.dtype}")
    xc = x if x.is_contiguous() else x.contiguous()
    N, Cc = xc.shape[0], xc.shape[1]
    S = xc.numel() // max(N * Cc, 1)
    y = torch.empty_like(xc)
    w = weight.detach().float().contiguous()
    b = bias.detach().float().contiguous()
    ws = _workspace(xc.device, L.mvi_groupnorm_workspace_bytes(N, Cc, S, num_groups))
    with torch.cuda.device(xc.device):
        _check(L.mvi_groupnorm_silu(xc.data_ptr(), y.data_ptr(), w.data_ptr(), b.data_ptr(), N, Cc, S, num_groups,
                                    float(eps), int(bool(silu)), _DT[x.dtype], ws.data_ptr(), ws.numel(),
                                    _stream(xc.device)), "group_norm")
    return y


def attention(q, k, v, heads):
    """q [B,Sq,H*D], k/v [B,Sk,H*D] -> [B,Sq,H*D]; scale = D**-0.5 (sgm/modules/attention.py:271)."""
    L = _lib.lib()
    if q.dtype not in _DT or k.dtype != q.dtype or v.dtype != q.dtype:
        raise TypeError(f"attention: q/k/v must share a dtype in {list(_DT)} (got {q.dtype}, {k.dtype}, {v.dtype})")
    B, Sq, HD = q.shape
    Sk = k.shape[1]
    D = HD // heads
    q, k, v = (t if t.is_contiguous() else t.contiguous() for t in (q, k, v))
    out = torch.empty_like(q)
    with torch.cuda.device(q.device):
        _check(L.mvi_attention_forward(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, heads, Sq, Sk, D,
                                       float(D) ** -0.5, _DT[q.dtype], _stream(q.device)), "attention")
    return out


def attention_kernel_kind(Sq, Sk, D, dtype):
    return int(_lib.lib().mvi_attention_kernel_kind(Sq, Sk, D, _DT[dtype]))
