"""GPU implementations of multiview_inpaint_amd.svd.ops through the C-ABI HIP library
(include/mvi_unet_ops.h). Importing this module without libmvi_hip.so raises."""
import ctypes as C
import math
import weakref

import torch

from .. import _lib

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
_ws = {}

# Optional per-op timing for bench.py: when PROFILE is a list, every op appends
# (kind, start_event, end_event, work) with events recorded on the stream the kernel runs on.
PROFILE = None
# When a list: every attention call appends (kernel variant, Sq, Sk) — 0 rowtile, 4 / 8 = the 4- / 8-wave MFMA kernel
# (mvi_attention_kernel_variant, the function the C dispatch itself uses). The parity tests assert from it WHICH kernel ran
# inside a module graph.
ATTN_VARIANTS = None


class _Timed:
    def __init__(self, kind, work, dev):
        self.on = PROFILE is not None
        if self.on:
            self.kind, self.work = kind, work
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.st = torch.cuda.current_stream(dev)

    def __enter__(self):
        if self.on:
            self.a.record(self.st)

    def __exit__(self, *exc):
        if self.on:
            self.b.record(self.st)
            PROFILE.append((self.kind, self.a, self.b, self.work))


def profile_summary():
    """{kind: (calls, total_ms, total_work)} of the recorded ops; call after torch.cuda.synchronize()."""
    out = {}
    for kind, a, b, work in PROFILE or []:
        c, ms, w = out.get(kind, (0, 0.0, 0.0))
        out[kind] = (c + 1, ms + a.elapsed_time(b), w + work)
    return out


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


_gn_sync = {}


def _groupnorm_sync(dev):
    """Zeroed once per (device, stream): the counters of the one-launch cluster GroupNorm (include/mvi_unet_ops.h,
    mvi_groupnorm_silu_ex2) — the kernels leave them zeroed, and launches that share a stream never overlap."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    b = _gn_sync.get(key)
    if b is None:
        b = torch.zeros(_lib.lib().mvi_groupnorm_sync_bytes(), dtype=torch.uint8, device=dev)
        _gn_sync[key] = b
    return b


def groupnorm_cluster_timeouts(dev=None):
    """Non-zero if a cluster GroupNorm block ever gave up waiting for its group on any stream of `dev` (sticky flag)."""
    tot = 0
    for (di, _), b in _gn_sync.items():
        if dev is None or di == torch.device(dev).index:
            tot += int(b.view(torch.int32)[2 * 65536].item())     # [arrived, departed] x 65536 groups, then the flag
    return tot


def check_groupnorm_cluster(dev=None):
    """Raises if a block of the one-launch cluster GroupNorm ever ran out of its bounded wait (the statistics of that call
    were then merged from stale partners). Cannot happen while the device makes progress (csrc/groupnorm_silu.hip: places are
    drawn as tickets in start order); called at the end of every SVDInpaintEngine.sample() and benchmark run."""
    n = groupnorm_cluster_timeouts(dev)
    if n:
        raise RuntimeError("cluster GroupNorm: a block gave up waiting for its group (device wedged?); results of this run "
                           "are not valid. MVI_GN_CLUSTER_KB=-1 selects the two-launch kernels.")


_f32_cache = {}


def _f32(p):
    """fp32 contiguous copy of a (small) parameter, cached per parameter object: the kernels take their affine
    parameters in fp32, the bf16-weights model stores them in bf16, and converting on every call costs a launch.
    The entry holds a weak reference to the parameter it was made from: Python reuses object ids and the caching
    allocator reuses addresses, so (id, data_ptr, version) of a freed model's parameter can all recur in the next
    model — a dead or different referent is a miss."""
    if p.dtype == torch.float32 and p.is_contiguous():
        return p.detach()
    key = (p.data_ptr(), p._version, p.dtype, tuple(p.shape), p.device)
    hit = _f32_cache.get(id(p))
    if hit is not None and hit[0] == key and hit[2]() is p:
        return hit[1]
    t = p.detach().float().contiguous()
    if len(_f32_cache) > 8192:
        _f32_cache.clear()
    _f32_cache[id(p)] = (key, t, weakref.ref(p))
    return t


def _gn(x, T, num_groups, weight, bias, eps, silu, chan_bias, stack3):
    L = _lib.lib()
    if x.dtype not in _DT:
        raise TypeError(f"group_norm: unsupported dtype {x.dtype}")
    xc = x if x.is_contiguous() else x.contiguous()
    N, Cc = xc.shape[0], xc.shape[1]
    S = xc.numel() // max(N * Cc, 1)
    y = torch.empty((N, 3 * Cc, *xc.shape[2:]) if stack3 else xc.shape, dtype=x.dtype, device=x.device)
    w, b = _f32(weight), _f32(bias)
    cb = None
    if chan_bias is not None:
        cb = chan_bias.detach().float().contiguous()
        if cb.shape != (N, Cc):
            raise ValueError(f"group_norm: chan_bias must be [{N}, {Cc}], got {tuple(cb.shape)}")
    ws = _workspace(xc.device, L.mvi_groupnorm_workspace_bytes(N, Cc, S, num_groups))
    sync = _groupnorm_sync(xc.device)
    with torch.cuda.device(xc.device), _Timed("groupnorm", (2.0 + 2.0 * bool(stack3)) * xc.numel() * xc.element_size(), xc.device):
        _check(L.mvi_groupnorm_silu_ex2(xc.data_ptr(), y.data_ptr(), w.data_ptr(), b.data_ptr(),
                                        None if cb is None else cb.data_ptr(), N // T, int(T), Cc, S, num_groups, float(eps),
                                        int(bool(silu)), int(bool(stack3)), _DT[x.dtype], ws.data_ptr(), ws.numel(),
                                        sync.data_ptr(), sync.numel(), _stream(xc.device)), "group_norm")
    return y


def group_norm_silu(x, num_groups, weight, bias, eps, silu, chan_bias=None):
    return _gn(x, 1, num_groups, weight, bias, eps, silu, chan_bias, False)


def group_norm_silu_tokens(x, num_groups, weight, bias, eps, silu, chan_bias=None):
    """GroupNorm(+SiLU) of x [N, C, *spatial] returned token-major [N, prod(spatial), C]."""
    L = _lib.lib()
    if x.dtype not in _DT:
        raise TypeError(f"group_norm: unsupported dtype {x.dtype}")
    xc = x if x.is_contiguous() else x.contiguous()
    N, Cc = xc.shape[0], xc.shape[1]
    S = xc.numel() // max(N * Cc, 1)
    y = torch.empty((N, S, Cc), dtype=x.dtype, device=x.device)
    cb = None
    if chan_bias is not None:
        cb = chan_bias.detach().float().contiguous()
        if cb.shape != (N, Cc):
            raise ValueError(f"group_norm: chan_bias must be [{N}, {Cc}], got {tuple(cb.shape)}")
    ws = _workspace(xc.device, L.mvi_groupnorm_workspace_bytes(N, Cc, S, num_groups))
    with torch.cuda.device(xc.device), _Timed("groupnorm_tokens", 2.0 * xc.numel() * xc.element_size(), xc.device):
        _check(L.mvi_groupnorm_silu_tokens(xc.data_ptr(), y.data_ptr(), _f32(weight).data_ptr(), _f32(bias).data_ptr(),
                                           None if cb is None else cb.data_ptr(), N, Cc, S, num_groups, float(eps),
                                           int(bool(silu)), _DT[x.dtype], ws.data_ptr(), ws.numel(), _stream(xc.device)),
               "group_norm_tokens")
    return y


def group_norm_silu_frames(x, T, num_groups, weight, bias, eps, silu, chan_bias=None, stack3=False):
    """x [(b T), C, *spatial] contiguous; statistics per (video, group) over all T frames."""
    return _gn(x, int(T), num_groups, weight, bias, eps, silu, chan_bias, stack3)


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
    kind = "attention_mfma" if L.mvi_attention_kernel_kind(Sq, Sk, D, _DT[q.dtype]) == 1 else "attention_rowtile"
    if ATTN_VARIANTS is not None:
        ATTN_VARIANTS.append((int(L.mvi_attention_kernel_variant(Sq, Sk, D, _DT[q.dtype])), Sq, Sk))
    with torch.cuda.device(q.device), _Timed(kind, 4.0 * B * heads * Sq * Sk * D, q.device):
        _check(L.mvi_attention_forward(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, heads, Sq, Sk, D,
                                       float(D) ** -0.5, _DT[q.dtype], _stream(q.device)), "attention")
    return out


def attention_packed(qkv, heads, q_log2=False):
    """Self-attention on ONE packed projection qkv [B, S, 3*H*D] = (q | k | v) -> [B, S, H*D]: the kernels read q, k, v in
    place with a token stride of 3*H*D (mvi_attention_forward_strided), so the three projections are one GEMM.
    q_log2: q carries D^-1/2 log2(e) from its projection's weights (mvi_attention_forward_strided_qlog2)."""
    L = _lib.lib()
    if qkv.dtype not in _DT or not qkv.is_contiguous():
        raise TypeError("attention_packed: contiguous fp32/bf16/f16 [B, S, 3*H*D] expected")
    B, S, C3 = qkv.shape
    HD = C3 // 3
    D = HD // heads
    out = torch.empty(B, S, HD, dtype=qkv.dtype, device=qkv.device)
    es = qkv.element_size()
    kind = "attention_mfma" if L.mvi_attention_kernel_kind(S, S, D, _DT[qkv.dtype]) == 1 else "attention_rowtile"
    if ATTN_VARIANTS is not None:
        ATTN_VARIANTS.append((int(L.mvi_attention_kernel_variant(S, S, D, _DT[qkv.dtype])), S, S))
    p = qkv.data_ptr()
    with torch.cuda.device(qkv.device), _Timed(kind, 4.0 * B * heads * S * S * D, qkv.device):
        if q_log2:
            rc = L.mvi_attention_forward_strided_qlog2(p, p + HD * es, p + 2 * HD * es, out.data_ptr(), B, heads, S, S, D, _DT[qkv.dtype],
                                                       C3, C3, HD, _stream(qkv.device))
        else:
            rc = L.mvi_attention_forward_strided(p, p + HD * es, p + 2 * HD * es, out.data_ptr(), B, heads, S, S, D,
                                                 float(D) ** -0.5, _DT[qkv.dtype], C3, C3, HD, _stream(qkv.device))
        _check(rc, "attention (packed)")
    return out


def attention_temporal_packed(qkv, heads, T, q_log2=False):
    """attention_temporal on a packed projection qkv [(bo*T), S, 3*H*D] -> [(bo*T), S, H*D]."""
    L = _lib.lib()
    if qkv.dtype not in _DT or not qkv.is_contiguous():
        raise TypeError("attention_temporal_packed: contiguous fp32/bf16/f16 [(bo T), S, 3*H*D] expected")
    BT, S, C3 = qkv.shape
    HD = C3 // 3
    D = HD // heads
    out = torch.empty(BT, S, HD, dtype=qkv.dtype, device=qkv.device)
    es = qkv.element_size()
    p = qkv.data_ptr()
    # HBM-bound (T = 14 keys per query): work = algorithmic bytes, q + k + v read and the output written once
    with torch.cuda.device(qkv.device), _Timed("attention_temporal", 4.0 * BT * S * HD * es, qkv.device):
        if q_log2:
            rc = L.mvi_attention_temporal_strided_qlog2(p, p + HD * es, p + 2 * HD * es, out.data_ptr(), BT // T, T, S, heads, D,
                                                        _DT[qkv.dtype], C3, HD, _stream(qkv.device))
        else:
            rc = L.mvi_attention_temporal_strided(p, p + HD * es, p + 2 * HD * es, out.data_ptr(), BT // T, T, S, heads, D,
                                                  float(D) ** -0.5, _DT[qkv.dtype], C3, HD, _stream(qkv.device))
        _check(rc, "attention_temporal (packed)")
    return out


def softmax_rows_(x, scale):
    """x [..., cols] contiguous: softmax(scale * x) over the last axis, in place."""
    L = _lib.lib()
    if x.dtype not in _DT or not x.is_contiguous():
        raise TypeError("softmax_rows_: contiguous fp32/bf16/f16 tensor expected")
    cols = x.shape[-1]
    rows = x.numel() // cols
    with torch.cuda.device(x.device), _Timed("softmax_rows", 2.0 * x.numel() * x.element_size(), x.device):
        _check(L.mvi_softmax_rows(x.data_ptr(), rows, cols, float(scale), _DT[x.dtype], _stream(x.device)), "softmax_rows")
    return x


_WIDE_SCORE_BYTES = 8 << 30        # scores held at once (of 288 GB)


def attention_wide(q, k, v):
    """Single-head attention with a wide head (D = C, e.g. 512 in the first-stage autoencoder's mid block): q [B,Sq,D],
    k/v [B,Sk,D] -> [B,Sq,D], scale D**-0.5. The D-deep contractions are library GEMMs (like the convolutions and
    Linear layers); the scores of a chunk of frames stay in HBM and are normalised in place by the HIP softmax."""
    B, Sq, D = q.shape
    Sk = k.shape[1]
    out = torch.empty_like(q)
    # Reduced-precision I/O (the opt-in bf16 / f16 first-stage decode): the scores and the probabilities stay fp32 — a D = 512 logit
    # rounded to bf16 before the softmax carries an error of |logit| 2^-9, which alone put the bf16 decode at 1.7 x the reference's own
    # bf16-autocast error (its scaled_dot_product_attention keeps the scores in fp32). 2 x 2 B Sq Sk D FLOPs on the fp32 matrix path:
    # ~25 ms of a 270 ms decode at 14 x 576x1024.
    wide = q.dtype != torch.float32
    per = Sq * Sk * (4 if wide else q.element_size())
    step = max(1, min(B, _WIDE_SCORE_BYTES // max(per, 1)))
    kt = k.transpose(1, 2)
    for b0 in range(0, B, step):
        if wide:
            s = torch.bmm(q[b0:b0 + step].float(), kt[b0:b0 + step].float())
            softmax_rows_(s, float(D) ** -0.5)
            out[b0:b0 + step] = torch.bmm(s, v[b0:b0 + step].float())
        else:
            s = torch.bmm(q[b0:b0 + step], kt[b0:b0 + step])
            softmax_rows_(s, float(D) ** -0.5)
            torch.bmm(s, v[b0:b0 + step], out=out[b0:b0 + step])
    return out


def attention_temporal(q, k, v, heads, T):
    """q/k/v [(bo*T), S, H*D] -> same; softmax over the T frames of each (video, token, head)."""
    L = _lib.lib()
    if q.dtype not in _DT or k.dtype != q.dtype or v.dtype != q.dtype:
        raise TypeError(f"attention_temporal: q/k/v must share a dtype in {list(_DT)}")
    BT, S, HD = q.shape
    D = HD // heads
    q, k, v = (t if t.is_contiguous() else t.contiguous() for t in (q, k, v))
    out = torch.empty_like(q)
    with torch.cuda.device(q.device), _Timed("attention_temporal", 4.0 * BT * S * HD * q.element_size(), q.device):
        _check(L.mvi_attention_temporal(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), BT // T, T, S, heads, D,
                                        float(D) ** -0.5, _DT[q.dtype], _stream(q.device)), "attention_temporal")
    return out


def geglu(h):
    L = _lib.lib()
    if h.dtype not in _DT:
        raise TypeError(f"geglu: unsupported dtype {h.dtype}")
    hc = h if h.is_contiguous() else h.contiguous()
    inner = hc.shape[-1] // 2
    rows = hc.numel() // (2 * inner)
    out = torch.empty(*hc.shape[:-1], inner, dtype=h.dtype, device=h.device)
    with torch.cuda.device(h.device), _Timed("geglu", 3.0 * rows * inner * h.element_size(), h.device):
        _check(L.mvi_geglu(hc.data_ptr(), out.data_ptr(), rows, inner, _DT[h.dtype], _stream(h.device)), "geglu")
    return out


def ff_geglu_supported(K, inner, dtype):
    return dtype in (torch.bfloat16, torch.float16) and bool(_lib.lib().mvi_ff_geglu_supported(int(K), int(inner), _DT[dtype]))


def ff_geglu(x, weight, bias):
    """GEGLU(x) = (x W_v^T + b_v) * gelu(x W_g^T + b_g) in one kernel: x [..., K], weight [2 inner, K], bias [2 inner] or None."""
    L = _lib.lib()
    K, inner = x.shape[-1], weight.shape[0] // 2
    xc = x.reshape(-1, K)
    if xc.stride(1) != 1 or xc.stride(0) % 8 or xc.data_ptr() % 16:      # the kernels read 16-byte vectors: a misaligned view is copied
        xc = xc.contiguous()
    wc = weight if weight.is_contiguous() and weight.data_ptr() % 16 == 0 else weight.contiguous().clone()
    rows = xc.shape[0]
    cap = int(L.mvi_ff_geglu_out_rows(rows))                     # whole 256-row blocks are stored
    full = torch.empty(cap, inner, dtype=x.dtype, device=x.device)
    out = full[:rows]
    b = None if bias is None else _f32(bias)
    # algorithmic bytes: x once, W once, out once; flops priced separately by the caller
    with torch.cuda.device(x.device), _Timed("ff_geglu", 4.0 * rows * K * inner, x.device):
        _check(L.mvi_ff_geglu(xc.data_ptr(), wc.data_ptr(), None if b is None else b.data_ptr(), full.data_ptr(), rows, cap, K, inner,
                              xc.stride(0), full.stride(0), _DT[x.dtype], _stream(x.device)), "ff_geglu")
    return out.reshape(*x.shape[:-1], inner)


def ff_geglu_n320_supported(K, inner, dtype):
    return dtype in (torch.bfloat16, torch.float16) and bool(_lib.lib().mvi_ff_geglu_n320_supported(int(K), int(inner), _DT[dtype]))


def ff_geglu_n320(x, weight, bias):
    """ff_geglu for a long contraction (K = 640 / 1280: the level-1 / level-2 FeedForward layers) on csrc/linear_n320.hip's GEGLU
    form (mvi_ff_geglu_n320): x [..., K], weight [2 inner, K], bias [2 inner] or None."""
    L = _lib.lib()
    K, inner = x.shape[-1], weight.shape[0] // 2
    xc = x.reshape(-1, K)
    if xc.stride(1) != 1 or xc.stride(0) % 8 or xc.data_ptr() % 16:
        xc = xc.contiguous()
    wc = weight if weight.is_contiguous() and weight.data_ptr() % 16 == 0 else weight.contiguous().clone()
    rows = xc.shape[0]
    cap = int(L.mvi_ff_geglu_out_rows(rows))
    full = torch.empty(cap, inner, dtype=x.dtype, device=x.device)
    b = None if bias is None else _f32(bias)
    with torch.cuda.device(x.device), _Timed("ff_geglu_n320", 4.0 * rows * K * inner, x.device):
        _check(L.mvi_ff_geglu_n320(xc.data_ptr(), wc.data_ptr(), None if b is None else b.data_ptr(), full.data_ptr(), rows, cap, K, inner,
                                   xc.stride(0), full.stride(0), _DT[x.dtype], _stream(x.device)), "ff_geglu_n320")
    return full[:rows].reshape(*x.shape[:-1], inner)


def linear_k320_supported(K, out_features, dtype):
    return dtype in (torch.bfloat16, torch.float16) and bool(_lib.lib().mvi_linear_k320_supported(int(K), int(out_features), _DT[dtype]))


def linear_k320(x, weight, bias):
    """F.linear(x, weight, bias) for K = 320 on the MFMA kernel of ff_geglu (plain epilogue): x [..., 320], weight [N, 320]."""
    L = _lib.lib()
    K, N = x.shape[-1], weight.shape[0]
    xc = x.reshape(-1, K)
    if xc.stride(1) != 1 or xc.stride(0) % 8 or xc.data_ptr() % 16:
        xc = xc.contiguous()
    wc = weight if weight.is_contiguous() and weight.data_ptr() % 16 == 0 else weight.contiguous().clone()
    rows = xc.shape[0]
    cap = int(L.mvi_ff_geglu_out_rows(rows))
    full = torch.empty(cap, N, dtype=x.dtype, device=x.device)
    b = None if bias is None else _f32(bias)
    with torch.cuda.device(x.device), _Timed("linear_k320", float(rows) * (K + N) * x.element_size(), x.device):
        _check(L.mvi_linear_k320(xc.data_ptr(), wc.data_ptr(), None if b is None else b.data_ptr(), full.data_ptr(), rows, cap, K, N,
                                 xc.stride(0), full.stride(0), _DT[x.dtype], _stream(x.device)), "linear_k320")
    return full[:rows].reshape(*x.shape[:-1], N)


def linear_n320_supported(K, out_features, dtype):
    return dtype in (torch.bfloat16, torch.float16) and bool(_lib.lib().mvi_linear_n320_supported(int(K), int(out_features), _DT[dtype]))


def linear_n320(x, weight, bias):
    """F.linear(x, weight, bias) for 320 outputs and K a multiple of 64 (csrc/linear_n320.hip): x [..., K], weight [320, K]."""
    L = _lib.lib()
    K, N = x.shape[-1], weight.shape[0]
    xc = x.reshape(-1, K)
    if xc.stride(1) != 1 or xc.stride(0) % 8 or xc.data_ptr() % 16:
        xc = xc.contiguous()
    wc = weight if weight.is_contiguous() else weight.contiguous()
    rows = xc.shape[0]
    cap = int(L.mvi_ff_geglu_out_rows(rows))
    full = torch.empty(cap, N, dtype=x.dtype, device=x.device)
    b = None if bias is None else _f32(bias)
    with torch.cuda.device(x.device), _Timed("linear_n320", 2.0 * rows * K * N, x.device):
        _check(L.mvi_linear_n320(xc.data_ptr(), wc.data_ptr(), None if b is None else b.data_ptr(), full.data_ptr(), rows, cap, K, N,
                                 xc.stride(0), full.stride(0), _DT[x.dtype], _stream(x.device)), "linear_n320")
    return full[:rows].reshape(*x.shape[:-1], N)


def linear_n320_add_layer_norm(x, weight, bias, ln_weight, ln_bias, eps, resid=None, row=None, ret_pre=False):
    """add_layer_norm(resid, h=F.linear(x, weight, bias), row=row) with the projection in the same kernel (mvi_linear_n320_add_layernorm):
    x [..., K], weight [320, K]; resid [..., 320] or None (then the sum starts from h); row [G, 320] / [G, 1, 320] or None.
    Returns (y, s, s_pre) like add_layer_norm: s the residual stream after the adds (h itself without resid and row), s_pre = resid + h
    only with ret_pre."""
    L = _lib.lib()
    K, N = x.shape[-1], weight.shape[0]
    xc = x.reshape(-1, K)
    if xc.stride(1) != 1 or xc.stride(0) % 8 or xc.data_ptr() % 16:
        xc = xc.contiguous()
    wc = weight if weight.is_contiguous() else weight.contiguous()
    rows = xc.shape[0]
    cap = int(L.mvi_ff_geglu_out_rows(rows))
    rc, rowc, row_div = None, None, 1
    if resid is not None:
        if resid.shape[-1] != N or resid.numel() != rows * N or resid.dtype != x.dtype:
            raise ValueError("linear_n320_add_layer_norm: resid must be [rows, 320] of x's dtype")
        rc = resid.reshape(rows, N)
        rc = rc if rc.is_contiguous() and rc.data_ptr() % 16 == 0 else rc.contiguous()
    if row is not None:
        if row.dtype != x.dtype or row.shape[-1] != N:
            raise ValueError("linear_n320_add_layer_norm: row must be [G, 320] / [G, 1, 320] of x's dtype")
        rowc = row.reshape(-1, N)
        rowc = rowc if rowc.is_contiguous() and rowc.data_ptr() % 16 == 0 else rowc.contiguous().clone()
        G = rowc.shape[0]
        if G == 0 or rows % G:
            raise ValueError(f"linear_n320_add_layer_norm: {rows} rows do not split into {G} equal runs")
        row_div = rows // G
    y = torch.empty(cap, N, dtype=x.dtype, device=x.device)
    s = torch.empty(cap, N, dtype=x.dtype, device=x.device)
    s_pre = torch.empty(cap, N, dtype=x.dtype, device=x.device) if (ret_pre and rowc is not None) else None
    b = None if bias is None else _f32(bias)
    # the 320 -> 320 projections (to_out, proj_in / proj_out: 53 GFLOP around ~0.7 GB of rows) are HBM-bound like linear_k320 and are
    # reported against that roof (VERDICT r5 "weak" 7): work = algorithmic bytes (x read; resid read; s, y and s_pre written)
    n_tensors = 2 + (rc is not None) + (s_pre is not None)
    kind, work = ("linear_n320_ln_k320", float(rows) * (K + N * n_tensors) * x.element_size()) if K <= N else ("linear_n320_ln", 2.0 * rows * K * N)
    with torch.cuda.device(x.device), _Timed(kind, work, x.device):
        _check(L.mvi_linear_n320_add_layernorm(xc.data_ptr(), wc.data_ptr(), None if b is None else b.data_ptr(), rows, cap, K, xc.stride(0),
                                               None if rc is None else rc.data_ptr(), None if rowc is None else rowc.data_ptr(), row_div,
                                               _f32(ln_weight).data_ptr(), _f32(ln_bias).data_ptr(), float(eps),
                                               None if s_pre is None else s_pre.data_ptr(), s.data_ptr(), y.data_ptr(), N, _DT[x.dtype],
                                               _stream(x.device)), "linear_n320_add_layer_norm")
    shape = (*x.shape[:-1], N)
    s = s[:rows].reshape(shape)
    if ret_pre:
        s_pre = s if s_pre is None else s_pre[:rows].reshape(shape)
    return y[:rows].reshape(shape), s, s_pre


def conv3x3_n320_supported(C_in, C_out, dtype):
    return dtype in (torch.bfloat16, torch.float16) and bool(_lib.lib().mvi_conv3x3_n320_supported(int(C_in), int(C_out), _DT[dtype]))


def conv3x3_n320_weight(weight):
    """conv.weight [C_out, C_in, 3, 3] in the kernel's order (mvi_conv3x3_n320_k_order): [C_out][C_in / 64][9 taps (ky, kx)][64
    channels] — or, order 0, [C_out][9][C_in], tap-major."""
    Co, Ci = weight.shape[0], weight.shape[1]
    taps = weight.permute(0, 2, 3, 1).reshape(Co, 9, Ci)                       # [Co, tap, c]
    if _lib.lib().mvi_conv3x3_n320_k_order(-1) == 1 and Ci % 64 == 0:
        return taps.reshape(Co, 9, Ci // 64, 64).permute(0, 2, 1, 3).reshape(Co, -1).contiguous()
    return taps.reshape(Co, -1).contiguous()


def conv3t_n320_weight(weight):
    """Conv3d weight [C_out, C_in, 3, 1, 1] in the kernel's order: [C_out][3 C_in], tap-major (kt, c)."""
    return weight[:, :, :, 0, 0].permute(0, 2, 1).reshape(weight.shape[0], -1).contiguous()


class GnPartials:
    """Statistics a producer kernel left for the GroupNorm behind it: `part` [samples * chunks * groups, 3] fp32 (count, mean, M2),
    `chunks` per sample, for `groups` groups and the per-sample channel bias `chan_bias` (the object, or None) they were taken with."""
    __slots__ = ("part", "chunks", "groups", "chan_bias")

    def __init__(self, part, chunks, groups, chan_bias):
        self.part, self.chunks, self.groups, self.chan_bias = part, int(chunks), int(groups), chan_bias


def conv_n320_gnstats_supported(rows, taps, C_in, C_out, spatial, groups, stride=1):
    return bool(_lib.lib().mvi_conv_n320_gnstats_supported(int(rows), int(taps), int(stride), int(C_in), int(C_out), int(spatial), int(groups)))


def _conv_taps_n320(kind, tok, weight_taps, bias, N, H, W, taps, stride=1, split=True, gn=None, up2=False):
    """tok: token-major activations of N images of H x W pixels (or, taps = 3, N videos of H frames of W pixels).
    gn = (groups, chan_bias [samples, C_out] fp32 or None): also leave the statistics of the GroupNorm that follows
    (mvi_conv3x3_n320_gnstats / mvi_conv3t_n320_gnstats) — returns (out, GnPartials); the caller asked
    conv_n320_gnstats_supported first."""
    L = _lib.lib()
    C = tok.shape[-1]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if up2:                                    # tok holds the H x W source of the 2x upsampled image the convolution runs over
        if taps != 9 or stride != 1 or gn is not None:
            raise ValueError(f"{kind}: upsampling goes with the plain 3x3 / stride 1 form")
        Ho, Wo = 2 * H, 2 * W
    rows = N * Ho * Wo
    if tok.numel() != N * H * W * C or weight_taps.shape[1] != taps * C or weight_taps.dtype != tok.dtype:
        raise ValueError(f"{kind}: token-major activations [.., C_in] of N H W rows and weight [C_out, {taps} C_in] of one dtype expected")
    xc = tok if tok.is_contiguous() and tok.data_ptr() % 16 == 0 else tok.contiguous().clone()
    wc = weight_taps if weight_taps.is_contiguous() else weight_taps.contiguous()
    Co = wc.shape[0]
    cap = int(L.mvi_ff_geglu_out_rows(rows))
    full = torch.empty(cap, Co, dtype=tok.dtype, device=tok.device)
    b = None if bias is None else _f32(bias)
    # > 0: a small image, K split over several blocks per tile
    ws_bytes = 0 if not split else int(L.mvi_conv3x3_n320_workspace_bytes(N, H, W, C, Co, stride) if taps == 9
                                       else L.mvi_conv3t_n320_workspace_bytes(N, H, W, C, Co))
    ws = _workspace(tok.device, ws_bytes) if ws_bytes else None
    wsp = None if ws is None else ws.data_ptr()
    if gn is not None:
        groups, cb = gn
        samples, spatial = (N, H * W) if taps == 9 else (N * H, W)
        if cb is not None and (cb.dtype != torch.float32 or not cb.is_contiguous() or tuple(cb.shape) != (samples, Co)):
            raise ValueError(f"{kind}: gn chan_bias must be contiguous fp32 [{samples}, {Co}]")
        nb = int(L.mvi_conv_n320_gnstats_bytes(samples, spatial, int(groups)))
        part = torch.empty(nb // 4, dtype=torch.float32, device=tok.device)
        fn = L.mvi_conv3x3_n320_gnstats if taps == 9 else L.mvi_conv3t_n320_gnstats
        with torch.cuda.device(tok.device), _Timed(kind, 2.0 * rows * taps * C * Co, tok.device):
            _check(fn(xc.data_ptr(), wc.data_ptr(), None if b is None else b.data_ptr(), full.data_ptr(), N, H, W, C, Co, cap, full.stride(0),
                      _DT[tok.dtype], None if cb is None else cb.data_ptr(), int(groups), part.data_ptr(), nb, _stream(tok.device)), kind + " (gn stats)")
        return full[:rows], GnPartials(part, spatial // 256, groups, cb)
    with torch.cuda.device(tok.device), _Timed(kind, 2.0 * rows * taps * C * Co, tok.device):
        if up2:
            rc = L.mvi_conv3x3_up2_n320(xc.data_ptr(), wc.data_ptr(), None if b is None else b.data_ptr(), full.data_ptr(), N, H, W, C, Co,
                                        cap, full.stride(0), _DT[tok.dtype], wsp, ws_bytes, _stream(tok.device))
        elif taps == 9:
            rc = L.mvi_conv3x3_n320(xc.data_ptr(), wc.data_ptr(), None if b is None else b.data_ptr(), full.data_ptr(), N, H, W, C, Co, stride,
                                    cap, full.stride(0), _DT[tok.dtype], wsp, ws_bytes, _stream(tok.device))
        else:
            rc = L.mvi_conv3t_n320(xc.data_ptr(), wc.data_ptr(), None if b is None else b.data_ptr(), full.data_ptr(), N, H, W, C, Co, cap,
                                   full.stride(0), _DT[tok.dtype], wsp, ws_bytes, _stream(tok.device))
        _check(rc, kind)
    return full[:rows]


def conv3x3_n320_fills_chip(N, H, W, C_in, C_out, min_blocks, stride=1):
    """Does a launch of this shape put at least min_blocks blocks on the chip (256 output pixels x 320 channels each, times the K
    split small images get)?"""
    rows = N * ((H - 1) // stride + 1) * ((W - 1) // stride + 1)
    blocks = -(-rows // 256) * (C_out // 320)
    return blocks >= min_blocks or int(_lib.lib().mvi_conv3x3_n320_workspace_bytes(N, H, W, C_in, C_out, stride)) > 0


def conv3t_n320_fills_chip(B, T, S, C_in, C_out, min_blocks):
    blocks = -(-B * T * S // 256) * (C_out // 320)
    return blocks >= min_blocks or int(_lib.lib().mvi_conv3t_n320_workspace_bytes(B, T, S, C_in, C_out)) > 0


def conv3x3_n320(tok, weight_taps, bias, H, W, stride=1, split=True, gn=None, up2=False):
    """3x3 / padding 1 convolution (stride 1 or 2) to a multiple of 320 output channels of token-major activations tok [N, H W, C_in]
    (csrc/linear_n320.hip in its implicit-GEMM mode) -> [N, Ho Wo, C_out]; weight_taps from conv3x3_n320_weight.
    gn = (groups, chan_bias): -> (out, GnPartials) for the GroupNorm that follows (see _conv_taps_n320).
    up2: the convolution of the nearest-neighbour 2x upsampled image (mvi_conv3x3_up2_n320) -> [N, (2 H)(2 W), C_out]."""
    N, S, C = tok.shape
    if S != H * W:
        raise ValueError("conv3x3_n320: tok [N, H W, C_in] expected")
    r = _conv_taps_n320("conv3x3_n320", tok, weight_taps, bias, N, H, W, 9, stride, split, gn=gn, up2=up2)
    if gn is not None:
        return r[0].view(N, -1, weight_taps.shape[0]), r[1]
    return r.view(N, -1, weight_taps.shape[0])


def conv3t_n320(tok, weight_taps, bias, T, split=True, gn=None):
    """(3, 1, 1) / padding (1, 0, 0) convolution over the frame axis of token-major activations tok [(b T), S, C_in] (frames of a video
    consecutive) -> [(b T), S, C_out]; weight_taps from conv3t_n320_weight. gn: as conv3x3_n320 (a sample = a frame)."""
    BT, S, C = tok.shape
    if BT % T:
        raise ValueError("conv3t_n320: tok [(b T), S, C_in] expected")
    r = _conv_taps_n320("conv3t_n320", tok, weight_taps, bias, BT // T, T, S, 3, 1, split, gn=gn)
    if gn is not None:
        return r[0].view(BT, S, -1), r[1]
    return r.view(BT, S, -1)


def stem_conv3x3_supported(conv, x):
    """A 3x3 / padding 1 Conv2d of the shapes csrc/stem_conv.hip builds (stride 1: <= 16 -> 16, <= 32 -> 32; stride 2: <= 16 -> 32) on a
    contiguous NCHW half tensor."""
    return (x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float16) and x.is_contiguous() and conv.weight.dtype == x.dtype
            and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) in ((1, 1), (2, 2)) and tuple(conv.padding) == (1, 1)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"
            and bool(_lib.lib().mvi_stem_conv3x3_supported(int(conv.in_channels), int(conv.out_channels), int(x.shape[3]), int(conv.stride[0]),
                                                           _DT[x.dtype])))


_stem_packed = {}


def _stem_packed_weight(weight, W, stride):
    """The weight in the kernel's fragment order (mvi_stem_conv3x3_pack), once per parameter version, width class and stride."""
    L = _lib.lib()
    key = (id(weight), int(stride))
    ver = (weight.data_ptr(), weight._version, weight.dtype, weight.device)
    hit = _stem_packed.get(key)
    if hit is None or hit[0]() is not weight or hit[1] != ver:
        Co, Ci = weight.shape[0], weight.shape[1]
        wc = weight.detach().contiguous()
        packed = torch.empty(int(L.mvi_stem_conv3x3_packed_bytes(Ci, Co, int(stride))), dtype=torch.uint8, device=weight.device)
        with torch.cuda.device(weight.device):
            _check(L.mvi_stem_conv3x3_pack(wc.data_ptr(), packed.data_ptr(), Ci, Co, int(W), int(stride), _DT[weight.dtype], _stream(weight.device)),
                   "stem_conv3x3_pack")
        hit = (weakref.ref(weight, lambda _r, k=key: _stem_packed.pop(k, None)), ver, packed)
        _stem_packed[key] = hit
    return hit[2]


def stem_conv3x3_silu(x, weight, bias, silu=True, stride=1):
    """silu(conv2d(x, weight, bias, stride, padding=1)) in one kernel; x [N, C_in, H, W], weight [C_out, C_in, 3, 3]."""
    L = _lib.lib()
    N, Cin, H, W = x.shape
    wc = _stem_packed_weight(weight, W, stride)
    b = None if bias is None else _f32(bias)
    y = torch.empty(N, weight.shape[0], (H - 1) // stride + 1, (W - 1) // stride + 1, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device), _Timed("stem_conv", float(x.numel() + y.numel()) * x.element_size(), x.device):
        _check(L.mvi_stem_conv3x3_silu(x.data_ptr(), wc.data_ptr(), None if b is None else b.data_ptr(), y.data_ptr(), N, Cin,
                                       weight.shape[0], H, W, int(stride), int(bool(silu)), _DT[x.dtype], _stream(x.device)), "stem_conv3x3_silu")
    return y


def bias_residual_add(h, bias, x):
    L = _lib.lib()
    if h.dtype not in _DT:
        raise TypeError(f"bias_residual_add: unsupported dtype {h.dtype}")
    hc = h if h.is_contiguous() else h.contiguous()
    xc = None
    if x is not None:
        if x.shape != h.shape or x.dtype != h.dtype:
            raise ValueError("bias_residual_add: x must match h in shape and dtype")
        xc = x if x.is_contiguous() else x.contiguous()
    N, Cc = hc.shape[0], hc.shape[1]
    S = hc.numel() // max(N * Cc, 1)
    b = None if bias is None else _f32(bias)
    out = torch.empty_like(hc)
    with torch.cuda.device(h.device), _Timed("bias_residual", (2.0 + (x is not None)) * hc.numel() * hc.element_size(), h.device):
        _check(L.mvi_bias_residual_add(hc.data_ptr(), None if xc is None else xc.data_ptr(), None if b is None else b.data_ptr(),
                                       out.data_ptr(), N, Cc, S, _DT[h.dtype], _stream(h.device)), "bias_residual_add")
    return out


def concat_add(h, skip, ctrl):
    """cat([h, skip + ctrl], dim=1) in one pass; ctrl may be None. All operands contiguous [N, C, *spatial], one dtype."""
    L = _lib.lib()
    if h.dtype not in _DT or skip.dtype != h.dtype or (ctrl is not None and ctrl.dtype != h.dtype):
        raise TypeError("concat_add: operands must share a supported dtype")
    if skip.shape[0] != h.shape[0] or skip.shape[2:] != h.shape[2:] or (ctrl is not None and ctrl.shape != skip.shape):
        raise ValueError("concat_add: shapes do not line up")
    if not (h.is_contiguous() and skip.is_contiguous() and (ctrl is None or ctrl.is_contiguous())):
        raise ValueError("concat_add: operands must be contiguous")
    N, C1, C2 = h.shape[0], h.shape[1], skip.shape[1]
    S = h.numel() // max(N * C1, 1) if C1 else skip.numel() // max(N * C2, 1)
    out = torch.empty((N, C1 + C2, *h.shape[2:]), dtype=h.dtype, device=h.device)
    nbytes = (2.0 * h.numel() + (2.0 + (ctrl is not None)) * skip.numel()) * h.element_size()
    with torch.cuda.device(h.device), _Timed("concat_add", nbytes, h.device):
        _check(L.mvi_concat_add(h.data_ptr(), skip.data_ptr(), None if ctrl is None else ctrl.data_ptr(), out.data_ptr(),
                                N, C1, C2, S, _DT[h.dtype], _stream(h.device)), "concat_add")
    return out


def bias_residual_blend(h, bias, x, alpha):
    """x + (1 - alpha[n]) * (h + bias[c]); alpha: [N] (any float dtype), one blend factor per sample of h [N, C, *spatial]."""
    L = _lib.lib()
    if h.dtype not in _DT:
        raise TypeError(f"bias_residual_blend: unsupported dtype {h.dtype}")
    if x.shape != h.shape or x.dtype != h.dtype:
        raise ValueError("bias_residual_blend: x must match h in shape and dtype")
    if alpha.numel() != h.shape[0]:
        raise ValueError(f"bias_residual_blend: alpha has {alpha.numel()} entries for {h.shape[0]} samples")
    hc = h if h.is_contiguous() else h.contiguous()
    xc = x if x.is_contiguous() else x.contiguous()
    N, Cc = hc.shape[0], hc.shape[1]
    S = hc.numel() // max(N * Cc, 1)
    b = None if bias is None else _f32(bias)
    a = alpha.detach().reshape(-1).float().contiguous()
    out = torch.empty_like(hc)
    with torch.cuda.device(h.device), _Timed("bias_residual", 3.0 * hc.numel() * hc.element_size(), h.device):
        _check(L.mvi_bias_residual_blend(hc.data_ptr(), xc.data_ptr(), None if b is None else b.data_ptr(), a.data_ptr(),
                                         out.data_ptr(), N, Cc, S, _DT[h.dtype], _stream(h.device)), "bias_residual_blend")
    return out


def bias_silu(h, bias):
    """silu(h + bias[c]) for h [N, C, *spatial], in place on h's (contiguous) memory."""
    L = _lib.lib()
    if h.dtype not in _DT:
        raise TypeError(f"bias_silu: unsupported dtype {h.dtype}")
    hc = h if h.is_contiguous() else h.contiguous()
    N, Cc = hc.shape[0], hc.shape[1]
    S = hc.numel() // max(N * Cc, 1)
    b = None if bias is None else _f32(bias)
    with torch.cuda.device(h.device), _Timed("bias_silu", 2.0 * hc.numel() * hc.element_size(), h.device):
        _check(L.mvi_bias_silu(hc.data_ptr(), None if b is None else b.data_ptr(), hc.data_ptr(), N, Cc, S, _DT[h.dtype],
                               _stream(h.device)), "bias_silu")
    return hc


def layernorm_supported(C_, dtype):
    return dtype in _DT and bool(_lib.lib().mvi_layernorm_supported(int(C_), _DT[dtype]))


def add_layer_norm(x, weight, bias, eps, h=None, row=None, ret_pre=False):
    """x [..., C] contiguous; returns (y, s, s_pre): s_pre = x + h, s = s_pre + row (broadcast over equal runs of
    rows), y = LayerNorm(s). s is None when neither h nor row is given; s_pre only when ret_pre and h."""
    L = _lib.lib()
    if x.dtype not in _DT:
        raise TypeError(f"add_layer_norm: unsupported dtype {x.dtype}")
    Cc = x.shape[-1]
    xc = x if x.is_contiguous() else x.contiguous()
    R = xc.numel() // Cc
    hc = rc = None
    row_div = 1
    if h is not None:
        if h.shape != x.shape or h.dtype != x.dtype:
            raise ValueError("add_layer_norm: h must match x in shape and dtype")
        hc = h if h.is_contiguous() else h.contiguous()
    if row is not None:
        if row.dtype != x.dtype or row.shape[-1] != Cc:
            raise ValueError("add_layer_norm: row must be [G, C] / [G, 1, C] of x's dtype")
        rc = row.reshape(-1, Cc)
        rc = rc if rc.is_contiguous() else rc.contiguous()
        G = rc.shape[0]
        if G == 0 or R % G:
            raise ValueError(f"add_layer_norm: {R} rows do not split into {G} equal runs")
        row_div = R // G
    y = torch.empty_like(xc)
    s = torch.empty_like(xc) if (hc is not None or rc is not None) else None
    s_pre = torch.empty_like(xc) if (ret_pre and hc is not None and rc is not None) else None
    n_io = 2 + (hc is not None) + (s is not None) + (s_pre is not None)
    with torch.cuda.device(x.device), _Timed("add_layernorm", float(n_io) * xc.numel() * xc.element_size(), x.device):
        _check(L.mvi_add_layernorm(xc.data_ptr(), None if hc is None else hc.data_ptr(), None if rc is None else rc.data_ptr(),
                                   row_div, _f32(weight).data_ptr(), _f32(bias).data_ptr(),
                                   None if s_pre is None else s_pre.data_ptr(), None if s is None else s.data_ptr(),
                                   y.data_ptr(), R, Cc, float(eps), _DT[x.dtype], _stream(x.device)), "add_layer_norm")
    if ret_pre and s_pre is None:                  # no separate buffer needed: s_pre coincides with x or with s
        s_pre = xc if hc is None else s
    return y, s, s_pre


def add_lerp(x, h, base, alpha):
    """lerp(x + h, base, alpha) with alpha [G] fp32 broadcast over equal runs of the rows of x [..., C]."""
    L = _lib.lib()
    if x.dtype not in _DT or base.dtype != x.dtype or base.shape != x.shape:
        raise TypeError("add_lerp: x and base must share shape and a supported dtype")
    Cc = x.shape[-1]
    xc = x if x.is_contiguous() else x.contiguous()
    bc = base if base.is_contiguous() else base.contiguous()
    hc = None
    if h is not None:
        if h.shape != x.shape or h.dtype != x.dtype:
            raise ValueError("add_lerp: h must match x in shape and dtype")
        hc = h if h.is_contiguous() else h.contiguous()
    R = xc.numel() // Cc
    al = alpha.detach().reshape(-1).float().contiguous()
    G = al.numel()
    if G == 0 or R % G:
        raise ValueError(f"add_lerp: {R} rows do not split into {G} equal runs")
    out = torch.empty_like(xc)
    with torch.cuda.device(x.device), _Timed("add_lerp", (3.0 + (hc is not None)) * xc.numel() * xc.element_size(), x.device):
        _check(L.mvi_add_lerp(xc.data_ptr(), None if hc is None else hc.data_ptr(), bc.data_ptr(), al.data_ptr(), R // G,
                              out.data_ptr(), R, Cc, _DT[x.dtype], _stream(x.device)), "add_lerp")
    return out


def tokens_to_planes_add(tok, x_in, bias=None, spatial=None):
    """tok [N, S, C] (+ bias[c]) + x_in [N, C, *spatial] -> [N, C, *spatial]; x_in None: the layout change alone, into `spatial`."""
    L = _lib.lib()
    if tok.dtype not in _DT or (x_in is not None and x_in.dtype != tok.dtype):
        raise TypeError("tokens_to_planes_add: tok and x_in must share a supported dtype")
    N, S, Cc = tok.shape
    if x_in is None:
        if spatial is None or math.prod(int(v) for v in spatial) != S:
            raise ValueError("tokens_to_planes_add: without x_in, `spatial` must be given and multiply to the token count")
        xc, out = None, torch.empty(N, Cc, *spatial, dtype=tok.dtype, device=tok.device)
    else:
        if x_in.shape[0] != N or x_in.shape[1] != Cc or x_in.numel() != tok.numel():
            raise ValueError(f"tokens_to_planes_add: tok {tuple(tok.shape)} does not match x_in {tuple(x_in.shape)}")
        xc = x_in if x_in.is_contiguous() else x_in.contiguous()
        out = torch.empty_like(xc)
    tc = tok if tok.is_contiguous() else tok.contiguous()
    xp = None if xc is None else xc.data_ptr()
    with torch.cuda.device(tok.device), _Timed("tokens_to_planes_add", (2.0 + (xc is not None)) * tc.numel() * tc.element_size(), tok.device):
        if bias is None:
            _check(L.mvi_tokens_to_planes_add(tc.data_ptr(), xp, out.data_ptr(), N, Cc, S, _DT[tok.dtype],
                                              _stream(tok.device)), "tokens_to_planes_add")
        else:
            _check(L.mvi_tokens_to_planes_add_bias(tc.data_ptr(), xp, _f32(bias).data_ptr(), out.data_ptr(), N, Cc, S,
                                                   _DT[tok.dtype], _stream(tok.device)), "tokens_to_planes_add")
    return out


def planes_add_to_tokens(x, tok, bias=None):
    """x [N, C, *spatial] + tok [N, S, C] (+ bias[c]) -> tokens [N, S, C]: a ResBlock's last add, result token-major."""
    L = _lib.lib()
    if x.dtype not in _DT or tok.dtype != x.dtype:
        raise TypeError("planes_add_to_tokens: x and tok must share a supported dtype")
    N, S, Cc = tok.shape
    if x.shape[0] != N or x.shape[1] != Cc or x.numel() != tok.numel():
        raise ValueError(f"planes_add_to_tokens: x {tuple(x.shape)} does not match tok {tuple(tok.shape)}")
    xc = x if x.is_contiguous() else x.contiguous()
    tc = tok if tok.is_contiguous() else tok.contiguous()
    out = torch.empty_like(tc)
    with torch.cuda.device(x.device), _Timed("planes_add_to_tokens", 3.0 * tc.numel() * tc.element_size(), x.device):
        _check(L.mvi_planes_add_to_tokens(xc.data_ptr(), tc.data_ptr(), None if bias is None else _f32(bias).data_ptr(), out.data_ptr(), N, Cc, S,
                                          _DT[x.dtype], _stream(x.device)), "planes_add_to_tokens")
    return out


def tokens_blend_to_planes(tok, base, bias, alpha, spatial):
    """base + (1 - alpha[n]) * (tok + bias[c]) for token-major tok, base [N, S, C] -> [N, C, *spatial]; alpha [N]."""
    L = _lib.lib()
    if tok.dtype not in _DT or base.dtype != tok.dtype or base.shape != tok.shape:
        raise TypeError("tokens_blend_to_planes: tok and base must share shape and a supported dtype")
    N, S, Cc = tok.shape
    if math.prod(int(v) for v in spatial) != S or alpha.numel() != N:
        raise ValueError("tokens_blend_to_planes: `spatial` must multiply to the token count and alpha hold one value per sample")
    tc = tok if tok.is_contiguous() else tok.contiguous()
    bc = base if base.is_contiguous() else base.contiguous()
    a = alpha.detach().reshape(N).float().contiguous()
    out = torch.empty(N, Cc, *spatial, dtype=tok.dtype, device=tok.device)
    with torch.cuda.device(tok.device), _Timed("tokens_blend_to_planes", 3.0 * tc.numel() * tc.element_size(), tok.device):
        _check(L.mvi_tokens_blend_to_planes(tc.data_ptr(), bc.data_ptr(), None if bias is None else _f32(bias).data_ptr(), a.data_ptr(),
                                            out.data_ptr(), N, Cc, S, _DT[tok.dtype], _stream(tok.device)), "tokens_blend_to_planes")
    return out


def planes_to_tokens(x, upsample=1):
    """x [N, C, H, W] -> tokens [N, H W, C]; upsample = 2: nearest-neighbour 2x upsampling folded in -> [N, (2H)(2W), C]."""
    L = _lib.lib()
    if x.dtype not in _DT or x.dim() != 4:
        raise TypeError("planes_to_tokens: [N, C, H, W] of a supported dtype expected")
    N, Cc, H, W = x.shape
    xc = x if x.is_contiguous() else x.contiguous()
    out = torch.empty(N, H * W * upsample * upsample, Cc, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device), _Timed("planes_to_tokens", float(xc.numel() + out.numel()) * x.element_size(), x.device):
        _check(L.mvi_planes_to_tokens(xc.data_ptr(), out.data_ptr(), N, Cc, H, W, int(upsample), _DT[x.dtype], _stream(x.device)),
               "planes_to_tokens")
    return out


def group_norm_silu_tok2tok(t, num_groups, weight, bias, eps, silu, chan_bias=None, frames=1, partials=None):
    """GroupNorm(+SiLU) of token-major t [N, S, C] -> [N, S, C] (csrc/groupnorm_tokens.hip); frames > 1: statistics over the `frames`
    consecutive samples of a video (the temporal ResBlock's norm), chan_bias still per sample. partials (GnPartials): the statistics
    the producer of t left (same groups, same chan_bias object) — no statistics pass (mvi_groupnorm_silu_tok2tok_pre)."""
    L = _lib.lib()
    if t.dtype not in _DT:
        raise TypeError(f"group_norm_tok2tok: unsupported dtype {t.dtype}")
    tc = t if t.is_contiguous() else t.contiguous()
    N, S, Cc = tc.shape
    nbytes = L.mvi_groupnorm_tok2tok_workspace_bytes(N, Cc, S, num_groups, _DT[t.dtype])
    if nbytes == 0:
        raise ValueError(f"group_norm_tok2tok: unsupported shape {tuple(t.shape)} / {num_groups} groups")
    cb = None
    if chan_bias is not None:
        cb = chan_bias.detach().float().contiguous()
        if cb.shape != (N, Cc):
            raise ValueError(f"group_norm_tok2tok: chan_bias must be [{N}, {Cc}], got {tuple(cb.shape)}")
    y = torch.empty_like(tc)
    ws = _workspace(tc.device, nbytes)
    if frames < 1 or N % frames:
        raise ValueError(f"group_norm_tok2tok: {N} samples are not whole videos of {frames} frames")
    if partials is not None:
        if partials.groups != num_groups or (partials.chan_bias is None) != (chan_bias is None) or \
                partials.part.numel() != N * partials.chunks * num_groups * 3:
            raise ValueError("group_norm_tok2tok: the producer's statistics do not belong to this norm")
        with torch.cuda.device(tc.device), _Timed("groupnorm_tok2tok", 2.0 * tc.numel() * tc.element_size(), tc.device):
            _check(L.mvi_groupnorm_silu_tok2tok_pre(tc.data_ptr(), y.data_ptr(), _f32(weight).data_ptr(), _f32(bias).data_ptr(),
                                                    None if cb is None else cb.data_ptr(), N, int(frames), Cc, S, num_groups, float(eps),
                                                    int(bool(silu)), _DT[t.dtype], partials.part.data_ptr(), partials.chunks,
                                                    ws.data_ptr(), ws.numel(), _stream(tc.device)), "group_norm_tok2tok (pre)")
        return y
    with torch.cuda.device(tc.device), _Timed("groupnorm_tok2tok", 2.0 * tc.numel() * tc.element_size(), tc.device):
        _check(L.mvi_groupnorm_silu_tok2tok_frames(tc.data_ptr(), y.data_ptr(), _f32(weight).data_ptr(), _f32(bias).data_ptr(),
                                                   None if cb is None else cb.data_ptr(), N, int(frames), Cc, S, num_groups, float(eps),
                                                   int(bool(silu)), _DT[t.dtype], ws.data_ptr(), ws.numel(), _stream(tc.device)),
               "group_norm_tok2tok")
    return y


def attention_kernel_variant(Sq, Sk, D, dtype):
    """0 = fp32-math rowtile kernel, 4 / 8 = the 4- / 8-wave MFMA kernel (the function the C dispatch itself uses)."""
    return int(_lib.lib().mvi_attention_kernel_variant(int(Sq), int(Sk), int(D), _DT[dtype]))


def attention_kernel_kind(Sq, Sk, D, dtype):
    return int(_lib.lib().mvi_attention_kernel_kind(Sq, Sk, D, _DT[dtype]))


# ---- Round 6: split-operand convolutions at fp32 accuracy (the first-stage decoder; csrc/linear_n320.hip, mvi_conv3x3_split3_f32) ------

def split_hi_lo(v):
    """fp32 -> (hi, lo) bf16 with hi + lo = v to 16 mantissa bits."""
    hi = v.to(torch.bfloat16)
    return hi, (v - hi.float()).to(torch.bfloat16)


_OPERANDS = {"split3": (3, torch.bfloat16, 0), "bf16": (1, torch.bfloat16, 1), "f16": (1, torch.float16, 2)}    # mode -> (terms, dtype, GN out_mode)


def split3_weight(weight, mode="split3"):
    """A convolution weight ([C_out, C_in, 3, 3] or [C_out, C_in, 3, 1, 1], any float type) for mvi_conv3x3_split3_f32 / mvi_conv3t_split3_f32:
    [C_out padded to whole column groups][taps x terms C_in] in the kernel's contraction order, padding rows zero. mode "split3": bf16
    with the logical channel axis (w_hi | w_lo | w_hi); "bf16" / "f16": the weight rounded once to that type (terms = 1)."""
    terms, dt, _ = _OPERANDS[mode]
    w = weight.detach().float()
    if terms == 3:
        hi, lo = split_hi_lo(w)
        w3 = torch.cat([hi, lo, hi], dim=1)
    else:
        w3 = w.to(dt)
    packed = conv3x3_n320_weight(w3) if w.dim() == 4 else conv3t_n320_weight(w3)
    Co = packed.shape[0]
    group = int(_lib.lib().mvi_conv_split3_group(Co))
    pad = -Co % group
    if pad:
        packed = torch.cat([packed, packed.new_zeros(pad, packed.shape[1])])
    return packed.contiguous()


def group_norm_split(x, num_groups, weight, bias, eps, silu, chan_bias=None, frames=1, mode="split3"):
    """GroupNorm(+SiLU) of fp32 token-major x [N, S, C] -> the operand of conv_split3 in `mode`: split bf16 [N, S, 2 C] = (hi | lo)
    ("split3") or one rounded value per element [N, S, C] ("bf16" / "f16") (mvi_groupnorm_silu_tok2tok_split).
    num_groups = 0: no normalisation, the plain split / rounding."""
    L = _lib.lib()
    terms, dt, out_mode = _OPERANDS[mode]
    if x.dtype != torch.float32 or x.dim() != 3:
        raise TypeError("group_norm_split: fp32 [N, S, C] expected")
    xc = x if x.is_contiguous() else x.contiguous()
    N, S, Cc = xc.shape
    y2 = torch.empty(N, S, (2 if terms == 3 else 1) * Cc, dtype=dt, device=x.device)
    ws, nbytes, cb = None, 0, None
    if num_groups:
        nbytes = L.mvi_groupnorm_tok2tok_workspace_bytes(N, Cc, S, num_groups, 0)
        if nbytes == 0:
            raise ValueError(f"group_norm_split: unsupported shape {tuple(x.shape)} / {num_groups} groups")
        ws = _workspace(x.device, nbytes)
        if chan_bias is not None:
            cb = chan_bias.detach().float().contiguous()
            if cb.shape != (N, Cc):
                raise ValueError(f"group_norm_split: chan_bias must be [{N}, {Cc}]")
    with torch.cuda.device(x.device), _Timed("groupnorm_split", 2.0 * xc.numel() * 4, x.device):
        _check(L.mvi_groupnorm_silu_tok2tok_split(xc.data_ptr(), y2.data_ptr(), None if not num_groups else _f32(weight).data_ptr(),
                                                  None if not num_groups else _f32(bias).data_ptr(), None if cb is None else cb.data_ptr(), N,
                                                  int(frames), Cc, S, int(num_groups), float(eps), int(bool(silu)), out_mode,
                                                  None if ws is None else ws.data_ptr(), nbytes, _stream(x.device)), "group_norm_split")
    return y2


def conv_split3(x2, w3, N, H, W, C_out, taps=9, mode="split3", _max_bytes=0xFFFFFFFF):
    """x . w at fp32 accuracy on the bf16 matrix pipe (mode "split3": x2 [N H W, 2 C] split bf16 from group_norm_split, w3 from
    split3_weight) — or with ONE rounded value per operand, fp32 accumulate (modes "bf16" / "f16": x2 [N H W, C]) -> fp32 [N H W, C_out],
    no bias. taps = 9: 3x3 / padding 1 over N images of H x W tokens; taps = 3: (3,1,1) / padding (1,0,0) over N videos of H frames of W
    tokens. The batch is cut so that every launch stays inside the kernel's 32-bit activation offsets."""
    L = _lib.lib()
    terms, dt, _ = _OPERANDS[mode]
    Cx = x2.shape[-1]
    C = Cx // 2 if terms == 3 else Cx
    rows = N * H * W
    if x2.dtype != dt or not x2.is_contiguous() or x2.numel() != rows * Cx or w3.shape[1] != taps * terms * C or w3.dtype != dt:
        raise ValueError(f"conv_split3 ({mode}): x2 contiguous {dt} [N H W, {'2 C' if terms == 3 else 'C'}] and w3 [C_out padded, taps {terms} C] expected")
    x2 = x2.reshape(rows, Cx)
    per = H * W * Cx * 2                                       # bytes of one image / video
    if per > _max_bytes and taps == 3 and W > 1:
        # one video alone exceeds the offsets (e.g. 25 frames of 576 x 1024 at 128 channels): the (3,1,1) convolution is independent
        # per pixel, so the pixel axis is cut (contiguous copies of the slices in, strided copies out: two extra passes, rare)
        parts = -(-per // _max_bytes)
        step = -(-W // parts)
        res = torch.empty(N, H, W, C_out, dtype=torch.float32, device=x2.device)
        xv = x2.view(N, H, W, Cx)
        for w0 in range(0, W, step):
            w1 = min(W, w0 + step)
            res[:, :, w0:w1] = conv_split3(xv[:, :, w0:w1].contiguous().view(-1, Cx), w3, N, H, w1 - w0, C_out, taps=3, mode=mode,
                                           _max_bytes=_max_bytes).view(N, H, w1 - w0, C_out)
        return res.view(rows, C_out)
    n_max = max(1, (_max_bytes // per))
    if per > 0xFFFFFFFF:
        raise ValueError("conv_split3: one image exceeds the kernel's 32-bit activation offsets")
    cap = int(L.mvi_conv_split3_out_rows(rows)) + 256
    out = torch.empty(cap, C_out, dtype=torch.float32, device=x2.device)
    fn = L.mvi_conv3x3_split3_f32 if taps == 9 else L.mvi_conv3t_split3_f32
    with torch.cuda.device(x2.device), _Timed("conv_split3", 2.0 * rows * taps * terms * C * C_out, x2.device):
        for n0 in range(0, N, n_max):
            n = min(n_max, N - n0)
            r0 = n0 * H * W
            _check(fn(x2[r0:].data_ptr(), w3.data_ptr(), out[r0:].data_ptr(), n, H, W, C, C_out, terms, _DT[dt], cap - r0, _stream(x2.device)),
                   "conv_split3")
    return out[:rows]


def rows_axpb(a, b, bias, alpha=1.0, out=None):
    """a + alpha * (b + bias[c]) for fp32 token-major a, b [..., C] in one pass (mvi_rows_axpb_f32); out may be a or b (default: b)."""
    L = _lib.lib()
    if a.dtype != torch.float32 or b.dtype != torch.float32 or a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous():
        raise TypeError("rows_axpb: contiguous fp32 tensors of one shape expected")
    out = b if out is None else out
    Cc = a.shape[-1]
    R = a.numel() // Cc
    bf = None if bias is None else _f32(bias)
    with torch.cuda.device(a.device), _Timed("rows_axpb", 3.0 * a.numel() * 4, a.device):
        _check(L.mvi_rows_axpb_f32(a.data_ptr(), b.data_ptr(), None if bf is None else bf.data_ptr(), float(alpha), out.data_ptr(), R, Cc,
                                   _stream(a.device)), "rows_axpb")
    return out


# ---- Round 6: block tails on token-major tensors with the next GroupNorm's statistics (csrc/groupnorm_tokens.hip gt_fused_kernel) -------

def rows_gnstats_supported(N, C, S, groups, dtype):
    return dtype in _DT and int(_lib.lib().mvi_rows_gnstats_bytes(int(N), int(C), int(S), int(groups), _DT[dtype])) > 0


def rows_fused(a, b=None, bias=None, base=None, alpha=None, groups=0, concat=False):
    """Token-major [N, S, C] tensors of one dtype. base None: a + b + bias[c] (b, bias optional) — the ResBlock's skip add, the
    transformer's `x + x_in`. base and alpha given: base + (1 - alpha[n]) * (a + bias[c]) — the temporal skip add + AlphaBlender (alpha [N]
    fp32). concat: channels (a | b + base) -> [N, S, Ca + Cb], base optional — the decoder's cat([h, skip + control], dim=1).
    groups > 0: also returns GnPartials of GroupNorm(groups) of the result (no chan_bias) for group_norm_silu_tok2tok(partials=...), None
    where the statistics kernel's geometry does not take the shape."""
    L = _lib.lib()
    if a.dim() != 3 or a.dtype not in _DT:
        raise TypeError("rows_fused: token-major [N, S, C] fp32 / bf16 / f16 expected")
    N, S, Ca = a.shape
    ts = [t for t in (a, b, base) if t is not None]
    if any(t.dtype != a.dtype or not t.is_contiguous() or t.shape[:2] != a.shape[:2] for t in ts):
        raise ValueError("rows_fused: a, b, base must be contiguous [N, S, .] tensors of one dtype")
    if concat:
        if b is None or bias is not None or alpha is not None or (base is not None and base.shape != b.shape):
            raise ValueError("rows_fused: concat takes a, b (and base shaped like b), no bias / alpha")
        mode, Cc = 2, Ca + b.shape[2]
    else:
        if any(t.shape != a.shape for t in ts):
            raise ValueError("rows_fused: a, b, base must share one shape")
        mode, Cc = (0 if base is None else 1), Ca
        if mode == 1 and (alpha is None or alpha.numel() != N):
            raise ValueError(f"rows_fused: alpha must be [{N}]")
    out = torch.empty(N, S, Cc, dtype=a.dtype, device=a.device)
    part, nb = None, 0
    if groups:
        nb = int(L.mvi_rows_gnstats_bytes(N, Cc, S, int(groups), _DT[a.dtype]))
        if nb == 0:
            groups = 0
        else:
            part = torch.empty(nb // 4, dtype=torch.float32, device=a.device)
    ch = C.c_int32(0)
    al = None if alpha is None else alpha.detach().float().contiguous()
    bf = None if bias is None else _f32(bias)
    kind = ("rows_add", "rows_blend", "rows_concat")[mode]
    nbytes = float(sum(t.numel() for t in ts) + out.numel()) * a.element_size()
    with torch.cuda.device(a.device), _Timed(kind, nbytes, a.device):
        _check(L.mvi_rows_fused_gnstats(mode, a.data_ptr(), None if b is None else b.data_ptr(), None if base is None else base.data_ptr(),
                                        None if bf is None else bf.data_ptr(), None if al is None else al.data_ptr(), out.data_ptr(), N, Cc, Ca, S,
                                        int(groups), _DT[a.dtype], None if part is None else part.data_ptr(), nb, C.byref(ch),
                                        _stream(a.device)), "rows_fused")
    if not groups:
        return out, None
    chunks = int(ch.value)
    return out, GnPartials(part[:N * chunks * int(groups) * 3], chunks, int(groups), None)
