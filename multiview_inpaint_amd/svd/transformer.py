"""Attention blocks of the video UNet: CrossAttention, GEGLU feed-forward, the spatial
BasicTransformerBlock / SpatialTransformer and the temporal VideoTransformerBlock /
SpatialVideoTransformer.

Reference: sgm/modules/attention.py:87-113 (GEGLU/FeedForward), :255-344 (CrossAttention),
:347-453 (MemoryEfficientCrossAttention), :456-572 (BasicTransformerBlock), :619-723
(SpatialTransformer); sgm/modules/video_attention.py:16-141 (VideoTransformerBlock), :147-302
(SpatialVideoTransformer). State-dict keys are identical to the reference.

Both attention-mode strings ("softmax", "softmax-xformers") map to the same CrossAttention, whose
softmax(QK^T/sqrt(d))V runs in ops.attention (HIP on the GPU). No mask is ever passed on the hot
path; a mask is rejected rather than ignored.
"""

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .layers import AlphaBlender, Tok, linear, maybe_checkpoint, timestep_embedding, to_tok, zero_module


import os

LOG2E = 1.4426950408889634
# MVI_ATTN_WEIGHT_FOLD=0: the packed projection keeps to_q.weight as it is and the kernels apply the softmax scale (same-box A/B runs)
FOLD_SCALE_INTO_WQ = os.environ.get("MVI_ATTN_WEIGHT_FOLD", "1") != "0"

# Cross-attention to the ONE CLIP token of the SVD conditioning (CrossAttention.single_token: `to_out(to_v(ctx))`, a row per image that the
# residual add broadcasts) was two 28-row GEMMs per layer: 92 launches of ~7.6 us per denoise step for 46 layers whose inputs are the same
# tensor. prepare_single_token_rows batches them per network call: the row is linear in the token, row = ctx (W_out W_v)^T + b_out, so ONE
# GEMM per kind against the layers' concatenated products W_out W_v (formed in fp32 once per parameter version, stored in the weights'
# type: one rounding of the matrix instead of one of the intermediate to_v(ctx) — the same size of error) gives every layer's row; a
# batched GEMM for the to_out's instead was 27 us per width and took the gain back (tools/experiments/token_rows_probe.py).
# MVI_SVD_BATCHED_TOKEN_ROWS=0: per layer, as rounds 2 - 5.
BATCHED_TOKEN_ROWS = os.environ.get("MVI_SVD_BATCHED_TOKEN_ROWS", "1") != "0"
CACHE_FRAME_MLP = os.environ.get("MVI_SVD_CACHE_FRAME_MLP", "1") != "0"       # 0: time_pos_embed evaluated in every call (same-box A/B runs)
_row_plans = {}               # id(root) -> (weakref(root), signature, {"s": plan, "t": plan})
_row_tables = []              # [(ctx, version, {id(attn2): row [n, 1, C]}, base context or None, (id(root), plan signature))], newest first


def _row_plan(root):
    """The cross-attention layers under `root` that meet the single context token — attn2 of every BasicTransformerBlock ("s": they see the
    per-frame context) and of every VideoTransformerBlock ("t": the first frame's token per video) — with, per kind, the layers' products
    W_out W_v concatenated (layers ordered by width), their to_out biases and where each width's rows sit. Rebuilt when a parameter changes."""
    import weakref
    kinds = {"s": [b.attn2 for b in root.modules() if isinstance(b, BasicTransformerBlock) and getattr(b, "attn2", None) is not None],
             "t": [v.attn2 for v in root.modules() if isinstance(v, VideoTransformerBlock) and getattr(v, "attn2", None) is not None]}

    def ok(a):
        o = a.to_out
        return (type(a.to_v) is nn.Linear and a.to_v.bias is None and isinstance(o, nn.Sequential) and len(o) == 2 and type(o[0]) is nn.Linear
                and o[0].bias is not None and isinstance(o[1], nn.Dropout) and not (o[1].training and o[1].p > 0)
                and o[0].in_features == a.to_v.out_features)
    kinds = {k: [a for a in v if ok(a)] for k, v in kinds.items()}
    params = [p for v in kinds.values() for a in v for p in (a.to_v.weight, a.to_out[0].weight, a.to_out[0].bias)]
    if len(params) < 12 or any(p.requires_grad and torch.is_grad_enabled() for p in params):
        return None
    sig = tuple((id(p), p._version, p.data_ptr()) for p in params)
    hit = _row_plans.get(id(root))
    if hit is not None and hit[0]() is root and hit[1] == sig:
        return hit[2]
    plans = {}
    with torch.no_grad():
        for k, mods in kinds.items():
            by_width = {}
            for a in mods:
                by_width.setdefault((a.to_out[0].out_features, a.to_v.out_features, a.to_v.in_features, a.to_v.weight.dtype, a.to_v.weight.device), []).append(a)
            if len({(key[2], key[3], key[4]) for key in by_width}) != 1:
                continue                                        # (one context width, dtype and device per kind, or the per-layer path)
            order = [a for key in by_width for a in by_width[key]]
            groups, off = [], 0
            for (C, _, _, _, _), ms in by_width.items():
                groups.append((C, off, [id(a) for a in ms]))
                off += C * len(ms)
            dt = order[0].to_v.weight.dtype
            M = torch.cat([(a.to_out[0].weight.float() @ a.to_v.weight.float()).to(dt) for a in order], 0).contiguous()     # [sum C, D]
            plans[k] = (M, torch.cat([a.to_out[0].bias for a in order], 0).contiguous(), groups)
    key = id(root)
    _row_plans[key] = (weakref.ref(root, lambda _r, kk=key: _row_plans.pop(kk, None)), sig, plans)
    return plans


def prepare_single_token_rows(root, context, T):
    """Fills this step's table of CrossAttention.single_token rows for every eligible layer of `root` (see BATCHED_TOKEN_ROWS): the
    spatial blocks' from `context` [N, 1, D], the temporal blocks' from the first frame's token of each video (context[::T],
    video_attention.py:250-254 — the object frame_context_of() hands to SpatialVideoTransformer.forward)."""
    if not (BATCHED_TOKEN_ROWS and torch.is_tensor(context) and context.is_cuda and context.dim() == 3 and context.shape[1] == 1) \
            or torch.is_grad_enabled() or context.requires_grad or not T or context.shape[0] % int(T):
        return
    plans = _row_plan(root)
    if not plans:
        return
    owner = (id(root), _row_plans[id(root)][1])
    for kind, ctx, base in (("s", context, None), ("t", frame_context_of(context, T), context)):
        plan = plans.get(kind)
        if plan is None:
            continue
        # the same context object, unchanged, met by the same network with the same parameters (the steps of a sample): its rows stand
        old = next((e for e in _row_tables if e[0] is ctx and e[1] == ctx._version and e[3] is base and e[4] == owner), None)
        if old is not None:
            continue
        M, b, groups = plan
        if M.dtype != ctx.dtype or M.shape[1] != ctx.shape[-1]:
            continue
        n = ctx.shape[0]
        R = F.linear(ctx.reshape(n, -1), M, b)                                       # [n, sum C]: every layer's to_out(to_v(ctx))
        table = {}
        for C, off, ids in groups:
            out = R[:, off:off + C * len(ids)].reshape(n, len(ids), C).transpose(0, 1).contiguous()    # [L, n, C]: a layer's rows contiguous
            for k, i in enumerate(ids):
                table[i] = out[k].unsqueeze(1)
        _row_tables.insert(0, (ctx, ctx._version, table, base, owner))
    del _row_tables[4:]                                          # (UNet + ControlNet, two kinds each; both key their temporal rows by ONE
                                                                 # frame-context object: frame_context_of returns the one in the table)


def frame_context_of(context, T):
    """context[::T] — as the very object the step's table of temporal rows was computed from, when there is one."""
    for c, ver, _, base, _ in _row_tables:
        if base is context and base._version == ver and c.shape[0] * int(T) == context.shape[0]:
            return c
    return context[::T]


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        return ops.linear_geglu(x, self.proj.weight, self.proj.bias)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.0):
        super().__init__()
        inner = int(dim * mult)
        first = GEGLU(dim, inner) if glu else nn.Sequential(nn.Linear(dim, inner), nn.GELU())
        self.net = nn.Sequential(first, nn.Dropout(dropout), nn.Linear(inner, dim if dim_out is None else dim_out))

    def forward(self, x, fuse=None):
        # (Dropout is the identity at inference; the output projection goes through ops.linear, which takes the MFMA kernel of
        # csrc/linear_n320.hip for the level-0 shape [258048, 1280] x [1280, 320] and is F.linear everywhere else)
        # fuse = {resid, norm, row, ret_pre}: the residual add(s) and the LayerNorm that follow this layer ride on its output projection
        # (ops.linear_add_layer_norm); the result is then add_layer_norm's (y, s, s_pre)
        if self.training and self.net[1].p > 0:
            h = self.net(x)
            return h if fuse is None else ops.finish_add_layer_norm(h, fuse)
        if fuse is not None:
            return ops.linear_add_layer_norm(self.net[0](x), self.net[2], fuse.get("resid"), fuse["norm"], row=fuse.get("row"),
                                             ret_pre=fuse.get("ret_pre", False))
        return ops.linear_module(self.net[2], self.net[0](x))


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0, backend=None, **_ignored):
        super().__init__()
        inner = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.scale, self.heads, self.dim_head = dim_head ** -0.5, heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))
        self.backend = backend

    def forward(self, x, context=None, mask=None, additional_tokens=None, n_times_crossframe_attn_in_self=0, fuse=None):
        """fuse = {resid, norm, row, ret_pre} (FeedForward.forward): the result is add_layer_norm's (y, s, s_pre) instead of the layer output."""
        if fuse is not None:
            if context is None and not n_times_crossframe_attn_in_self and additional_tokens is None and mask is None \
                    and ops.packed_ok(x, self.heads, self.dim_head):
                w, q_log2 = self._packed_qkv_weight(x.dtype, fold=ops.attention_scale_fold_pays(x, self.dim_head))
                return ops.linear_add_layer_norm(ops.attention_packed(ops.linear(x, w), self.heads, q_log2=q_log2), self.to_out,
                                                 fuse.get("resid"), fuse["norm"], row=fuse.get("row"), ret_pre=fuse.get("ret_pre", False))
            return ops.finish_add_layer_norm(self.forward(x, context, mask, additional_tokens, n_times_crossframe_attn_in_self), fuse)
        if mask is not None:
            raise NotImplementedError("attention masks are not part of the SVD denoise path")
        n_extra = 0
        if additional_tokens is not None:
            n_extra = additional_tokens.shape[1]
            x = torch.cat([additional_tokens, x], dim=1)
        ctx = x if context is None else context
        if ctx.shape[1] == 1 and not n_times_crossframe_attn_in_self and not n_extra:
            # one key: softmax over a single score is exactly 1, so every query receives the value
            # row; to_q / to_k never influence the result (SURVEY.md §7 "S_k = 1 cross-attention":
            # 32 of the 64 attention calls of a UNet eval attend to the single CLIP token). The output
            # projection is applied to that one row and the result broadcast over the queries (the
            # residual add broadcasts it) — also keeps zero-stride operands out of the GEMM library.
            return self.to_out(self.to_v(ctx)).expand(-1, x.shape[1], -1)
        if context is None and not n_times_crossframe_attn_in_self and not n_extra and ops.packed_ok(x, self.heads, self.dim_head):
            # self-attention at inference: one GEMM [.., C] x [C, 3 H D] instead of three passes over the activations;
            # the attention kernel reads q, k, v out of the packed result in place
            # the softmax scale rides in the q weights exactly where it buys speed: sequences the 8-wave kernel serves
            w, q_log2 = self._packed_qkv_weight(x.dtype, fold=ops.attention_scale_fold_pays(x, self.dim_head))
            return ops.linear_module(self.to_out, ops.attention_packed(ops.linear(x, w), self.heads, q_log2=q_log2))
        q, k, v = self.to_q(x), self.to_k(ctx), self.to_v(ctx)
        if n_times_crossframe_attn_in_self:
            n = n_times_crossframe_attn_in_self
            assert x.shape[0] % n == 0
            k = k[::n].repeat_interleave(x.shape[0] // n, dim=0)
            v = v[::n].repeat_interleave(x.shape[0] // n, dim=0)
        out = ops.attention(q, k, v, self.heads)
        if n_extra:
            out = out[:, n_extra:]
        return self.to_out(out)


    def _packed_qkv_weight(self, act_dtype=None, fold=False):
        """(cat(to_q.weight, to_k.weight, to_v.weight) [3 H D, C], q_log2), rebuilt only when a weight changes (inference weights
        are static; not a parameter or buffer, so the state-dict keys stay the reference's).
        fold (reduced precision on the GPU, FOLD_SCALE_INTO_WQ; asked for by the spatial self-attentions the 8-wave MFMA kernel
        serves, S >= 1024): the q rows are round(dim_head^-1/2 * log2(e) * to_q.weight), the product taken in fp32 and rounded
        once to the weights' type, so the packed projection's q already is the exponent of 2 the softmax needs (attention.py:332-336
        `softmax(q k^T * scale)`) and the kernel's softmax loses its one multiply per score — the port that bounds it. Price: the
        320 .. 1280 weights behind a q element are rounded a second time (the in-kernel alternative rounds q itself a second
        time: the same size of perturbation); measured on the op (tests/test_unet_ops_gpu.py::test_softmax_scale_folded_...) and
        inside the reference-pinned graphs at full size. Two cached forms per module at most (folded / plain)."""
        ws = (self.to_q.weight, self.to_k.weight, self.to_v.weight)
        fold = bool(fold and FOLD_SCALE_INTO_WQ and ws[0].is_cuda and ws[0].dtype in (torch.bfloat16, torch.float16)
                    and (act_dtype is None or act_dtype == ws[0].dtype) and not torch.is_grad_enabled())
        key = tuple((w.data_ptr(), w._version, w.dtype, w.device) for w in ws)
        slot = "_wqkv_folded" if fold else "_wqkv"
        hit = getattr(self, slot, None)
        if hit is None or hit[0] != key:
            wq = ws[0].detach()
            if fold:
                wq = (wq.float() * (self.scale * LOG2E)).to(wq.dtype)
            hit = (key, torch.cat([wq, ws[1].detach(), ws[2].detach()], dim=0).contiguous(), fold)
            setattr(self, slot, hit)
        return hit[1], hit[2]

    def forward_temporal(self, x, T, fuse=None):
        """Self-attention over frames for x [(b t), s, c] in place of regroup -> forward -> regroup back (fuse: see forward)."""
        if ops.packed_ok(x, self.heads, self.dim_head) and self.to_k.in_features == self.to_q.in_features:
            w, q_log2 = self._packed_qkv_weight(x.dtype)      # (plain: T keys per softmax, nothing to gain from the fold)
            o = ops.attention_temporal_packed(ops.linear(x, w), self.heads, T, q_log2=q_log2)
            if fuse is not None:
                return ops.linear_add_layer_norm(o, self.to_out, fuse.get("resid"), fuse["norm"], row=fuse.get("row"),
                                                 ret_pre=fuse.get("ret_pre", False))
            return ops.linear_module(self.to_out, o)
        h = self.to_out(ops.attention_temporal(self.to_q(x), self.to_k(x), self.to_v(x), self.heads, T))
        return h if fuse is None else ops.finish_add_layer_norm(h, fuse)

    def single_token(self, ctx):
        """Cross-attention to ONE context token: the projected value row (see forward) — from this step's batched table when the
        network prepared one (prepare_single_token_rows), else two small GEMMs here."""
        for c, ver, table, _, _ in _row_tables:                 # (UNet and ControlNet see the same context: each has its own table)
            if c is ctx and ver == ctx._version:
                hit = table.get(id(self))
                if hit is not None:
                    return hit
        return self.to_out(self.to_v(ctx))


# the xformers-backed class of the reference; same maths, same parameters
MemoryEfficientCrossAttention = CrossAttention


class BasicTransformerBlock(nn.Module):
    ATTENTION_MODES = {"softmax": CrossAttention, "softmax-xformers": MemoryEfficientCrossAttention}

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False, attn_mode="softmax", sdp_backend=None):
        super().__init__()
        assert attn_mode in self.ATTENTION_MODES
        cls = self.ATTENTION_MODES[attn_mode]
        self.disable_self_attn = disable_self_attn
        self.attn1 = cls(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                         context_dim=context_dim if disable_self_attn else None, backend=sdp_backend)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = cls(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                         backend=sdp_backend)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.checkpoint = checkpoint

    def forward(self, x, context=None, additional_tokens=None, n_times_crossframe_attn_in_self=0):
        if additional_tokens is None and not n_times_crossframe_attn_in_self:
            return maybe_checkpoint(self._forward, self.checkpoint, x, context)
        return self._forward(x, context, additional_tokens, n_times_crossframe_attn_in_self)

    def _forward(self, x, context=None, additional_tokens=None, n_times_crossframe_attn_in_self=0):
        if additional_tokens is not None or n_times_crossframe_attn_in_self:
            x = self.attn1(self.norm1(x), context=context if self.disable_self_attn else None,
                           additional_tokens=additional_tokens,
                           n_times_crossframe_attn_in_self=0 if self.disable_self_attn else n_times_crossframe_attn_in_self) + x
            x = self.attn2(self.norm2(x), context=context, additional_tokens=additional_tokens) + x
            return self.ff(self.norm3(x)) + x
        h, skip = self.forward_deferred(x, context)
        return h + skip

    def forward_deferred(self, x, context=None, n1=None, ff_fuse=None):
        """The block with its last residual add left to the caller: returns (h, skip), block output = h + skip.
        Every inner residual add is fused with the LayerNorm that follows it — and both ride on the output projection of the layer
        that produced the addend (CrossAttention / FeedForward `fuse`: one kernel at the level-0 width, linear + add_layer_norm
        elsewhere). With a single context token the cross-attention output is one row per image (CrossAttention.forward), norm2
        cannot influence it and is not evaluated, and both inner adds collapse into one pass.
        n1: norm1(x) when the caller already has it (the stem's proj_in carries it). ff_fuse = {norm, row, ret_pre}: the caller's NEXT
        add + LayerNorm (the temporal block's entry) rides on this block's FeedForward; the return value is then
        add_layer_norm(skip, norm, h=h, row=row, ret_pre=ret_pre)'s (y, s, s_pre) instead of (h, skip)."""
        if n1 is None:
            n1, _, _ = ops.add_layer_norm(x, self.norm1)
        ctx = x if context is None else context
        self_ctx = context if self.disable_self_attn else None
        if context is not None and ctx.shape[1] == 1:
            n3, x, _ = self.attn1(n1, context=self_ctx, fuse=dict(resid=x, norm=self.norm3, row=self.attn2.single_token(ctx)))
        else:
            n2, x, _ = self.attn1(n1, context=self_ctx, fuse=dict(resid=x, norm=self.norm2))
            n3, x, _ = self.attn2(n2, context=context, fuse=dict(resid=x, norm=self.norm3))
        if ff_fuse is not None:
            return self.ff(n3, fuse=dict(ff_fuse, resid=x))
        return self.ff(n3), x


def Normalize(in_channels):
    """GroupNorm(32, eps=1e-6) of the transformer stem (attention.py:125-128) — not followed by SiLU."""
    from .layers import GroupNorm32
    return GroupNorm32(32, in_channels, eps=1e-6, affine=True)


class SpatialTransformer(nn.Module):
    """GN -> proj_in -> depth x BasicTransformerBlock over the h*w tokens -> proj_out -> + input."""
    takes = "attn"

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None,
                 disable_self_attn=False, use_linear=False, attn_type="softmax", use_checkpoint=True,
                 sdp_backend=None):
        super().__init__()
        if context_dim is not None and not isinstance(context_dim, (list, tuple)):
            context_dim = [context_dim]
        if context_dim is None:
            context_dim = [None] * depth
        elif len(context_dim) != depth:
            assert all(c == context_dim[0] for c in context_dim), "need homogenous context_dim to match depth automatically"
            context_dim = depth * [context_dim[0]]
        self.in_channels, self.use_linear = in_channels, use_linear
        inner = n_heads * d_head
        self.norm = Normalize(in_channels)
        self.proj_in = nn.Linear(in_channels, inner) if use_linear else nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim[d],
                                  disable_self_attn=disable_self_attn, attn_mode=attn_type,
                                  checkpoint=use_checkpoint, sdp_backend=sdp_backend) for d in range(depth)])
        self.proj_out = zero_module(nn.Linear(inner, in_channels) if use_linear else nn.Conv2d(inner, in_channels, 1))

    takes_tokens = True

    def _norm_tokens(self, x):
        """self.norm(x) as tokens b (h w) c: from planes the norm writes them directly; from the token-major stream (layers.Tok) it is
        the token norm, without its statistics pass where the block before left them."""
        if isinstance(x, Tok):
            n = self.norm
            return ops.group_norm_tok2tok(x.t, n.num_groups, n.weight, n.bias, n.eps, silu=False, partials=x.gn_stats(n.num_groups))
        return self.norm.forward_tokens(x)

    def _tok_route_ok(self, x):
        from .layers import _tok2tok_ok
        N, C, H, W = x.shape
        return self.use_linear and _tok2tok_ok(N, C, H * W, self.norm.num_groups, x.dtype)

    def _tokens_in(self, x):
        if self.use_linear:
            return ops.linear_module(self.proj_in, self._norm_tokens(x))
        h = self.proj_in(self.norm(x))
        return h.flatten(2).transpose(1, 2).contiguous()      # b c h w -> b (h w) c

    def _tokens_out(self, t, x_in):
        b, c, h, w = x_in.shape
        if isinstance(x_in, Tok):
            # `x + x_in` on token rows, with the statistics of the next block's first norm
            from . import hip_ops
            from .layers import GN_STATS_FROM_TAILS
            out, st = hip_ops.rows_fused(ops.linear_module(self.proj_out, t), x_in.t, groups=self.norm.num_groups if GN_STATS_FROM_TAILS else 0)
            return Tok(out, h, w, st)
        if self.use_linear:
            return ops.tokens_to_planes_add(ops.linear_module(self.proj_out, t), x_in)   # b (h w) c -> b c h w, + x_in, one pass
        t = t.transpose(1, 2).reshape(b, -1, h, w)
        return self.proj_out(t) + x_in

    def forward(self, x, context=None):
        if isinstance(x, Tok) and not self._tok_route_ok(x):
            return to_tok(self.forward(x.planes(), context))
        ctxs = context if isinstance(context, list) else [context]
        t = self._tokens_in(x)
        for i, blk in enumerate(self.transformer_blocks):
            t = blk(t, context=ctxs[0 if len(ctxs) == 1 else i])
        return self._tokens_out(t, x)


class VideoTransformerBlock(nn.Module):
    """Transformer block over the time axis: tokens are regrouped `(b t) s c -> (b s) t c`
    (video_attention.py:110-141)."""
    ATTENTION_MODES = {"softmax": CrossAttention, "softmax-xformers": MemoryEfficientCrossAttention}

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True,
                 timesteps=None, ff_in=False, inner_dim=None, attn_mode="softmax", disable_self_attn=False,
                 disable_temporal_crossattention=False, switch_temporal_ca_to_sa=False):
        super().__init__()
        cls = self.ATTENTION_MODES[attn_mode]
        self.ff_in = ff_in or inner_dim is not None
        inner_dim = dim if inner_dim is None else inner_dim
        assert int(n_heads * d_head) == inner_dim
        self.is_res = inner_dim == dim
        if self.ff_in:
            self.norm_in = nn.LayerNorm(dim)
            self.ff_in = FeedForward(dim, dim_out=inner_dim, dropout=dropout, glu=gated_ff)
        self.timesteps, self.disable_self_attn = timesteps, disable_self_attn
        if disable_self_attn:
            self.attn1 = cls(query_dim=inner_dim, heads=n_heads, dim_head=d_head, context_dim=context_dim, dropout=dropout)
        else:
            self.attn1 = cls(query_dim=inner_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(inner_dim, dim_out=dim, dropout=dropout, glu=gated_ff)
        if disable_temporal_crossattention:
            if switch_temporal_ca_to_sa:
                raise ValueError
            self.attn2 = None
        else:
            self.norm2 = nn.LayerNorm(inner_dim)
            if switch_temporal_ca_to_sa:
                self.attn2 = cls(query_dim=inner_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
            else:
                self.attn2 = cls(query_dim=inner_dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.norm1, self.norm3 = nn.LayerNorm(inner_dim), nn.LayerNorm(inner_dim)
        self.switch_temporal_ca_to_sa, self.checkpoint = switch_temporal_ca_to_sa, checkpoint

    def forward(self, x, context=None, timesteps=None):
        return maybe_checkpoint(lambda a, c: self._forward(a, c, timesteps), self.checkpoint, x, context)

    def forward_in_place_layout(self, h, skip, emb, frame_context, timesteps, entry=None):
        """Same block on tokens [(b t), s, c] WITHOUT regrouping to (b s) t c: LayerNorm, the feed-forwards
        and the projections are per-token, the frame self-attention reads its T rows with a stride
        (ops.attention_temporal), and the cross-attention sees one context token per video (`frame_context`
        [b, 1, ctx_dim], the first frame's token: video_attention.py:250-254), so its result is one row per
        video broadcast over frames and positions (norm2 cannot influence it and is not evaluated).

        Input: the spatial block's output still split as h + skip (BasicTransformerBlock.forward_deferred) and
        the frame-index embedding `emb` [(b t), 1, c]. Returns (x_spatial, f, x): x_spatial = h + skip, the
        block output is f + x (x None when the block is not residual) — left un-added for the blend.
        Every residual / broadcast add rides on the LayerNorm that follows it."""
        assert not self.disable_self_attn and not self.switch_temporal_ca_to_sa
        T = int(self.timesteps or timesteps)
        # entry: (norm(s), s, x_spatial) with x_spatial = h + skip, s = x_spatial + emb — from the spatial block's FeedForward when the
        # caller had it carry them (`entry`: BasicTransformerBlock.forward_deferred(ff_fuse=...) with self.entry_norm)
        if entry is None:
            entry = ops.add_layer_norm(skip, self.entry_norm, h=h, row=emb, ret_pre=True)
        if self.ff_in:
            ni, x, x_spatial = entry
            n1, x, _ = self.ff_in(ni, fuse=dict(resid=x if self.is_res else None, norm=self.norm1))
        else:
            n1, x, x_spatial = entry
        row = self.attn2.single_token(frame_context) if self.attn2 is not None else None     # [b, 1, c]
        n3, x, _ = self.attn1.forward_temporal(n1, T, fuse=dict(resid=x, norm=self.norm3, row=row))
        return x_spatial, self.ff(n3), (x if self.is_res else None)

    @property
    def entry_norm(self):
        """The LayerNorm the block's input meets first (norm_in in front of ff_in, else norm1)."""
        return self.norm_in if self.ff_in else self.norm1

    def _forward(self, x, context=None, timesteps=None):
        assert self.timesteps or timesteps
        assert not (self.timesteps and timesteps) or self.timesteps == timesteps
        t = int(self.timesteps or timesteps)
        B, S, C = x.shape
        x = x.reshape(B // t, t, S, C).transpose(1, 2).reshape(-1, t, C)      # (b t) s c -> (b s) t c
        if self.ff_in:
            skip = x
            x = self.ff_in(self.norm_in(x))
            if self.is_res:
                x = x + skip
        x = self.attn1(self.norm1(x), context=context if self.disable_self_attn else None) + x
        if self.attn2 is not None:
            x = self.attn2(self.norm2(x), context=None if self.switch_temporal_ca_to_sa else context) + x
        skip = x
        x = self.ff(self.norm3(x))
        if self.is_res:
            x = x + skip
        return x.reshape(B // t, S, t, C).transpose(1, 2).reshape(B, S, C)    # (b s) t c -> (b t) s c

    def get_last_layer(self):
        return self.ff.net[-1].weight


class SpatialVideoTransformer(SpatialTransformer):
    """Per depth: spatial block, + frame-index embedding, temporal block, alpha blend
    (video_attention.py:147-302)."""
    takes = "video_attn"

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, use_linear=False, context_dim=None,
                 use_spatial_context=False, timesteps=None, merge_strategy="fixed", merge_factor=0.5,
                 time_context_dim=None, ff_in=False, checkpoint=False, time_depth=1, attn_mode="softmax",
                 disable_self_attn=False, disable_temporal_crossattention=False, max_time_embed_period=10000):
        super().__init__(in_channels, n_heads, d_head, depth=depth, dropout=dropout, attn_type=attn_mode,
                         use_checkpoint=checkpoint, context_dim=context_dim, use_linear=use_linear,
                         disable_self_attn=disable_self_attn)
        self.time_depth, self.depth, self.max_time_embed_period = time_depth, depth, max_time_embed_period
        inner = n_heads * d_head
        if use_spatial_context:
            time_context_dim = context_dim
        self.time_stack = nn.ModuleList([
            VideoTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=time_context_dim,
                                  timesteps=timesteps, checkpoint=checkpoint, ff_in=ff_in, inner_dim=inner,
                                  attn_mode=attn_mode, disable_self_attn=disable_self_attn,
                                  disable_temporal_crossattention=disable_temporal_crossattention)
            for _ in range(depth)])
        assert len(self.time_stack) == len(self.transformer_blocks)
        self.use_spatial_context = use_spatial_context
        self.time_pos_embed = nn.Sequential(linear(in_channels, in_channels * 4), nn.SiLU(),
                                            linear(in_channels * 4, in_channels))
        self.time_mixer = AlphaBlender(alpha=merge_factor, merge_strategy=merge_strategy)

    def _frame_embedding(self, T, b, device):
        """Sinusoidal embedding of the frame index, [b*T, in_channels] in the MLP's dtype. It depends only on the
        shape, so it is built once per (T, b, device, dtype) instead of with eight small kernels per call."""
        wd = self.time_pos_embed[0].weight.dtype
        key = (T, b, str(device), wd)
        cache = self.__dict__.setdefault("_pe_cache", {})
        pe = cache.get(key)
        if pe is None:
            frame_idx = torch.arange(T, device=device).repeat(b)
            pe = timestep_embedding(frame_idx, self.in_channels, repeat_only=False, max_period=self.max_time_embed_period)
            pe = cache[key] = pe.to(wd)
        return pe

    def _frame_embedding_mlp(self, T, b, device):
        """time_pos_embed(frame-index embedding)[:, None, :] (video_attention.py:256-266): its input depends on the shape only, so outside
        autograd the MLP's output is kept until one of its parameters changes (two small GEMMs + SiLU per transformer and call: 69
        launches per denoise step for 23 tensors that never change between steps)."""
        pe = self._frame_embedding(T, b, device)
        mlp = self.time_pos_embed
        if not CACHE_FRAME_MLP or (torch.is_grad_enabled() and any(p.requires_grad for p in mlp.parameters())):
            return mlp(pe)[:, None, :]
        key = (id(pe),) + tuple((p.data_ptr(), p._version) for p in mlp.parameters())
        hit = self.__dict__.get("_pe_mlp")
        if hit is None or hit[0] != key or hit[1] is not pe:
            with torch.no_grad():
                hit = (key, pe, mlp(pe)[:, None, :])
            self.__dict__["_pe_mlp"] = hit
        return hit[2]

    def forward(self, x, context=None, time_context=None, timesteps=None, image_only_indicator=None):
        if isinstance(x, Tok) and not self._tok_route_ok(x):
            return to_tok(self.forward(x.planes(), context, time_context, timesteps, image_only_indicator))
        _, _, h, w = x.shape
        T = int(timesteps)
        # SVD configuration (use_spatial_context, one CLIP token): the temporal blocks run on the
        # spatial token layout, no "(b t) s c <-> (b s) t c" copies (4 full-tensor moves per block)
        in_place = (self.use_spatial_context and context is not None and context.ndim == 3 and context.shape[1] == 1
                    and all(not m.disable_self_attn and not m.switch_temporal_ca_to_sa for m in self.time_stack))
        if in_place:
            frame_context = frame_context_of(context, T)                       # [b, 1, ctx_dim]
        elif self.use_spatial_context:
            assert context.ndim == 3, f"n dims of spatial context should be 3 but are {context.ndim}"
            time_context = context[::T].repeat_interleave(h * w, dim=0)        # first frame's context per pixel
        elif time_context is not None:
            time_context = time_context.repeat_interleave(h * w, dim=0)
            if time_context.ndim == 2:
                time_context = time_context[:, None]
        n1 = None
        if in_place and self.use_linear:
            # the stem's projection carries the first block's norm1 (one kernel at the level-0 width)
            n1, t, _ = ops.linear_add_layer_norm(self._norm_tokens(x), self.proj_in, None, self.transformer_blocks[0].norm1)
        else:
            t = self._tokens_in(x)
        emb = self._frame_embedding_mlp(T, x.shape[0] // T, x.device)
        alpha = None
        for blk, mix in zip(self.transformer_blocks, self.time_stack):
            if in_place:
                # the spatial block's FeedForward carries the temporal block's entry: + skip, + frame embedding, LayerNorm
                entry = blk.forward_deferred(t, context, n1=n1, ff_fuse=dict(norm=mix.entry_norm, row=emb, ret_pre=True))
                n1 = None
                x_spatial, f, x_t = mix.forward_in_place_layout(None, None, emb, frame_context, T, entry=entry)
                if alpha is None:
                    alpha = self.time_mixer.get_alpha(image_only_indicator)
                    if alpha.numel() > 1 and alpha.size(0) != t.size(0):
                        alpha = self.time_mixer.get_alpha(image_only_indicator, rows=t.size(0))   # the reference's CFG patch (util.py:365-367)
                    # alpha.to(x.dtype) of the reference (util.py:369), kept as the fp32 vector add_lerp takes: converted once per alpha
                    # object (get_alpha hands out the same tensor while the indicator stands) instead of twice per call (two launches)
                    hit = self.__dict__.get("_alpha_rows")
                    if hit is None or hit[0] is not alpha or hit[1] != t.dtype or torch.is_grad_enabled():
                        rows_alpha = alpha.reshape(-1).to(t.dtype)
                        hit = (alpha, t.dtype, rows_alpha if torch.is_grad_enabled() else rows_alpha.float())
                        if not torch.is_grad_enabled():
                            self.__dict__["_alpha_rows"] = hit
                    alpha = hit[2]
                # alpha * spatial + (1 - alpha) * (f + x_t), with the temporal block's last residual add inside
                t = ops.add_lerp(f if x_t is None else x_t, None if x_t is None else f, x_spatial, alpha)
            else:
                t = blk(t, context=context)
                tt = mix(t + emb, context=time_context, timesteps=T)
                t = self.time_mixer(x_spatial=t, x_temporal=tt, image_only_indicator=image_only_indicator)
        return self._tokens_out(t, x)
