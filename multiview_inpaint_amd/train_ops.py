"""Host side of include/mvi_train_ops.h: the ops either side of the rasterizer inside the timed region of the 3DGS
training loop, with the reference's own function names so that a training script changes one import
line (`from utils.loss_utils import l1_loss, ssim` -> `from multiview_inpaint_amd.train_ops import l1_loss, ssim`;
gs-simp/train.py:17, gs-simp/inpaint_rec.py:16).

GPU tensors run the HIP kernels through the C-ABI and raise if the library is missing; there is no CPU path here
(the CPU restatement lives in oracle/loss_oracle.py and is test infrastructure)."""
import ctypes as C
import weakref

import torch

from . import _lib

_ws = {}


def _check(rc, what):
    if rc != 0:
        msg = _lib.lib().mvi_train_last_error().decode(errors="replace")
        raise (ValueError if rc == -1 else RuntimeError)(f"{what} failed ({rc}): {msg}")


def _workspace(dev, nbytes):
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _ws[key] = w
    return w


def _prep(image, gt, weight):
    if not image.is_cuda:
        raise RuntimeError(f"photometric loss: tensors must be on the GPU (got {image.device}); there is no CPU path")
    if image.ndim != 3 or image.shape[0] != 3 or gt.shape != image.shape:
        raise ValueError(f"photometric loss: image and gt must be [3,H,W] (got {tuple(image.shape)}, {tuple(gt.shape)})")
    H, W = image.shape[1:]
    img = image.detach().to(torch.float32).contiguous()
    g = gt.detach().to(device=image.device, dtype=torch.float32).contiguous()
    w = None
    if weight is not None:
        if weight.numel() != H * W:
            raise ValueError(f"photometric loss: weight must have H*W = {H * W} elements (got {tuple(weight.shape)})")
        w = weight.detach().to(device=image.device, dtype=torch.float32).reshape(H, W).contiguous()
    return img, g, w, H, W


def photometric_loss_forward_backward(image, gt, lambda_dssim=0.2, weight=None, need_grad=True, upstream=1.0):
    """One fused evaluation. Returns (out3, grad): out3 = device tensor [loss, mean|x-y|, mean SSIM] (no host sync),
    grad = upstream * d loss / d image [3,H,W] or None."""
    L = _lib.lib()
    img, g, w, H, W = _prep(image, gt, weight)
    dev = img.device
    out3 = torch.empty(3, dtype=torch.float32, device=dev)
    grad = torch.empty_like(img) if need_grad else None
    ws = _workspace(dev, L.mvi_photometric_loss_workspace_bytes(H, W))
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        _check(L.mvi_photometric_loss(p(img), p(g), p(w), H, W, float(lambda_dssim), float(upstream), p(out3), p(grad),
                                      p(ws), ws.numel(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
               "photometric_loss")
    return out3, grad


class _PhotometricLoss(torch.autograd.Function):
    """which: 0 = combined loss, 1 = mean|x-y|, 2 = mean SSIM (lambda fixed to select the gradient)."""

    @staticmethod
    def forward(ctx, image, gt, weight, lambda_dssim, which):
        lam = {0: lambda_dssim, 1: 0.0, 2: 1.0}[which]
        out3, grad = photometric_loss_forward_backward(image, gt, lam, weight, need_grad=image.requires_grad)
        ctx.save_for_backward(grad)
        ctx.sign = -1.0 if which == 2 else 1.0          # d(1 - ssim) = -d ssim: the kernel differentiates the LOSS
        ctx.in_dtype = image.dtype
        return out3[which].clone()

    @staticmethod
    def backward(ctx, g_out):
        (grad,) = ctx.saved_tensors
        if grad is None:
            return None, None, None, None, None
        return (grad * (g_out * ctx.sign)).to(ctx.in_dtype), None, None, None, None


def fused_l1_dssim_loss(image, gt, lambda_dssim=0.2, mask=None):
    """(1 - lambda) * l1_loss(x, y) + lambda * (1 - ssim(x, y)), x = image * (1 - mask), y = gt * (1 - mask) —
    gs-simp/train.py:91-92; with `mask` = gt_mask [1,H,W] the non-inpainted-view branch of
    gs-simp/inpaint_rec.py:120-123. Differentiable with respect to `image`."""
    weight = None if mask is None else 1.0 - mask.to(torch.float32)
    return _PhotometricLoss.apply(image, gt, weight, float(lambda_dssim), 0)


class _LossPair(torch.autograd.Function):
    """(mean|x - y|, mean SSIM) of one image pair as ONE autograd node: the statistics pass runs once for both values, and the backward
    is ONE gradient pass weighted by the two upstream gradients (device scalars, never read back) — mvi_photometric_loss_stats /
    mvi_photometric_loss_grad2. The derivative maps live in a buffer of the node's own until the backward."""

    @staticmethod
    def forward(ctx, image, gt):
        L = _lib.lib()
        img, g, _, H, W = _prep(image, gt, None)
        dev = img.device
        need = image.requires_grad
        nbytes = L.mvi_photometric_loss_workspace_bytes(H, W)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if need else _workspace(dev, nbytes)
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            _check(L.mvi_photometric_loss_stats(p(img), p(g), None, H, W, p(out3), p(ws), ws.numel(),
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "photometric_loss_stats")
        if need:
            ctx.save_for_backward(img, g, ws)
        ctx.in_dtype = image.dtype
        return out3[1].clone(), out3[2].clone()

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        global _pending_pair
        if _pending_pair is not None and _pending_pair[-1] is ctx:
            _pending_pair = None                          # this node's buffers are released with this backward: nothing may join it later
        if not ctx.saved_tensors:
            return None, None
        img, g, ws = ctx.saved_tensors
        L = _lib.lib()
        dev = img.device
        z = torch.zeros((), dtype=torch.float32, device=dev)
        w2 = torch.stack([z if g_l1 is None else g_l1.to(torch.float32).reshape(()),
                          z if g_ssim is None else g_ssim.to(torch.float32).reshape(())])
        grad = torch.empty_like(img)
        p = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            _check(L.mvi_photometric_loss_grad2(p(img), p(g), None, img.shape[1], img.shape[2], p(w2), p(grad), p(ws), ws.numel(),
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "photometric_loss_grad2")
        return grad.to(ctx.in_dtype), None


# l1_loss(a, b) followed by ssim(a, b) on the SAME tensor objects (the reference's loss expression, train.py:91-92): the second call
# returns the other output of the first call's node. Matched by object identity and in-place version (weak references to a and b), by
# the autograd state both calls ran under (grad mode and a.requires_grad: an l1_loss under no_grad must not hand a detached SSIM to
# a differentiated ssim()), and only while the node has not run its backward (which frees its buffers: the slot is cleared there).
# The SSIM half waits in this one slot until the matching ssim() takes it or the next l1_loss() replaces it — the caller may have
# dropped the L1 value by then (`0.8 * l1_loss(a, b) + 0.2 * (1 - ssim(a, b))` in one expression) — so at most one node is pinned by it.
_pending_pair = None


def _autograd_state(t):
    return (torch.is_grad_enabled(), bool(t.requires_grad))


def l1_loss(network_output, gt):
    """gs-simp/utils/loss_utils.py:17-18 (same name and signature)."""
    global _pending_pair
    l1, ss = _LossPair.apply(network_output, gt)         # (raises on bad shapes / devices before anything is remembered)
    _pending_pair = (weakref.ref(network_output), network_output._version, weakref.ref(gt), gt._version, ss,
                     _autograd_state(network_output), ss.grad_fn)
    return l1


def ssim(img1, img2, window_size=11, size_average=True):
    """gs-simp/utils/loss_utils.py:33-41 (same name and signature; the window size used by every caller is 11)."""
    global _pending_pair
    if window_size != 11 or not size_average:
        raise NotImplementedError("ssim: only window_size=11, size_average=True (the only form the training scripts use)")
    pair, _pending_pair = _pending_pair, None
    if (pair is not None and pair[0]() is img1 and pair[1] == img1._version and pair[2]() is img2 and pair[3] == img2._version
            and pair[5] == _autograd_state(img1)):
        return pair[4]
    return _PhotometricLoss.apply(img1, img2, None, 1.0, 2)


# ---------------------------------------------------------------------------------------------------------------------
# Adam

class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps) — weight_decay 0, amsgrad off: the configuration of
    gs-simp/scene/gaussian_model.py:163 — stepping every parameter tensor of all groups in ONE HIP launch.
    State layout is torch.optim.Adam's (state[p] = {"step", "exp_avg", "exp_avg_sq"}), so the reference's
    densification code, which edits optimizer.state directly (gaussian_model.py:335-404), and state_dict()
    checkpoints (gaussian_model.py:61-93 capture/restore) work unchanged. Replace
    `torch.optim.Adam(l, lr=0.0, eps=1e-15)` by `FusedAdam(l, lr=0.0, eps=1e-15)`."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        # tensors that share (betas, eps, step) go into one launch; the reference has one such set
        buckets = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise RuntimeError("FusedAdam: fp32 GPU parameters only; there is no CPU path")
                if p.grad.is_sparse or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: dense contiguous parameters only")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                key = (group["betas"], group["eps"], int(st["step"]), p.device)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not (st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()):
                    st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].contiguous(), st["exp_avg_sq"].contiguous()
                buckets.setdefault(key, []).append((p, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"])))
        for (betas, eps, step, dev), items in buckets.items():
            for i in range(0, len(items), 8):
                chunk = items[i:i + 8]
                arr = (_lib.AdamGroup * len(chunk))()
                for k, (p, g, m, v, lr) in enumerate(chunk):
                    arr[k] = _lib.AdamGroup(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr)
                with torch.cuda.device(dev):
                    _check(L.mvi_adam_step(arr, len(chunk), float(betas[0]), float(betas[1]), float(eps), step,
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "adam_step")
        return loss


# ---------------------------------------------------------------------------------------------------------------------
# parameter activations

class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_scaling, raw_rotation, raw_opacity, features_dc, features_rest):
        L = _lib.lib()
        ts = [raw_scaling, raw_rotation, raw_opacity, features_dc, features_rest]
        if not all(t.is_cuda and t.dtype == torch.float32 for t in ts):
            raise RuntimeError("activate_gaussians: fp32 GPU tensors only; there is no CPU path")
        rs, rr, ro, fd, fr = (t.detach().contiguous() for t in ts)
        P, M = rs.shape[0], 1 + fr.shape[1]
        if rs.shape != (P, 3) or rr.shape != (P, 4) or ro.numel() != P or fd.shape != (P, 1, 3) or fr.shape != (P, M - 1, 3):
            raise ValueError("activate_gaussians: expected [P,3], [P,4], [P,1], [P,1,3], [P,M-1,3]")
        dev = rs.device
        f32 = dict(dtype=torch.float32, device=dev)
        scales, rots, opac = torch.empty(P, 3, **f32), torch.empty(P, 4, **f32), torch.empty(P, 1, **f32)
        shs = torch.empty(P, M, 3, **f32)
        p = lambda t: None if t.numel() == 0 else C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            _check(L.mvi_gaussian_activations(P, M, p(rs), p(rr), p(ro), p(fd), p(fr), p(scales), p(rots), p(opac), p(shs),
                                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "gaussian_activations")
        ctx.save_for_backward(rr, scales, opac)
        ctx.M = M
        return scales, rots, opac, shs

    @staticmethod
    def backward(ctx, d_scales, d_rots, d_opac, d_shs):
        L = _lib.lib()
        rr, scales, opac = ctx.saved_tensors
        P, M, dev = rr.shape[0], ctx.M, rr.device
        f32 = dict(dtype=torch.float32, device=dev)
        z = lambda g, shape: torch.zeros(shape, **f32) if g is None else g.to(torch.float32).contiguous()
        d_scales, d_rots, d_opac, d_shs = z(d_scales, (P, 3)), z(d_rots, (P, 4)), z(d_opac, (P, 1)), z(d_shs, (P, M, 3))
        o_s, o_r, o_o = torch.empty(P, 3, **f32), torch.empty(P, 4, **f32), torch.empty(P, 1, **f32)
        o_dc, o_rest = torch.empty(P, 1, 3, **f32), torch.empty(P, M - 1, 3, **f32)
        p = lambda t: None if t.numel() == 0 else C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            _check(L.mvi_gaussian_activations_backward(P, M, p(rr), p(scales), p(opac), p(d_scales), p(d_rots), p(d_opac),
                                                       p(d_shs), p(o_s), p(o_r), p(o_o), p(o_dc), p(o_rest),
                                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   "gaussian_activations_backward")
        return o_s, o_r, o_o, o_dc, o_rest


def activate_gaussians(raw_scaling, raw_rotation, raw_opacity, features_dc, features_rest):
    """(get_scaling, get_rotation, get_opacity, get_features) of gs-simp/scene/gaussian_model.py:95-115 in one launch,
    differentiable: returns scales [P,3], rotations [P,4], opacities [P,1], shs [P,M,3]."""
    return _Activate.apply(raw_scaling, raw_rotation, raw_opacity, features_dc, features_rest)


# ---------------------------------------------------------------------------------------------------------------------
# simple_knn

def distCUDA2(points):
    """simple_knn._C.distCUDA2: [N,3] fp32 GPU tensor -> [N] mean squared distance to the 3 nearest other points
    (gs-simp/scene/gaussian_model.py:134, :546, :623). Same name and signature as the plug-in."""
    if not points.is_cuda:
        raise RuntimeError(f"distCUDA2: points must be on the GPU (got {points.device}); there is no CPU path")
    if points.ndim != 2 or points.shape[1] != 3:
        raise ValueError(f"distCUDA2: points must be [N,3] (got {tuple(points.shape)})")
    p = points.detach().to(torch.float32).contiguous()
    out = torch.empty(p.shape[0], dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        _check(_lib.lib().mvi_knn3_mean_dist2(C.c_void_p(p.data_ptr()) if p.numel() else None, p.shape[0],
                                              C.c_void_p(out.data_ptr()) if p.numel() else None,
                                              C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)), "distCUDA2")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# prune_points / _prune_optimizer tensor surgery

def compact_rows(keep_mask, tensors):
    """[t[keep_mask] for t in tensors] for tensors sharing their first dimension P — the boolean indexing of
    prune_points / _prune_optimizer (gs-simp/scene/gaussian_model.py:351-382) over the 6 parameters, 12 Adam moments and
    3 statistics — with ONE scan of the mask and ONE gather launch (bit-identical results). 4-byte element types."""
    tensors = list(tensors)
    if not keep_mask.is_cuda:
        raise RuntimeError("compact_rows: GPU tensors expected; there is no CPU path")
    P = keep_mask.shape[0]
    if keep_mask.ndim != 1 or any(t.shape[0] != P or t.device != keep_mask.device for t in tensors):
        raise ValueError("compact_rows: mask [P] and tensors [P, ...] on one device expected")
    if any(t.element_size() != 4 for t in tensors):
        raise TypeError("compact_rows: 4-byte element types only")
    L, dev = _lib.lib(), keep_mask.device
    m = (keep_mask if keep_mask.dtype in (torch.bool, torch.uint8) else keep_mask != 0).contiguous()
    srcs = [t.detach().contiguous() for t in tensors]
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        nbytes = L.mvi_compact_workspace_bytes(P)
        ws = _workspace(dev, nbytes + 256)
        cnt = ws[nbytes:nbytes + 4].view(torch.int32)         # the kept count lives behind the plan
        _check(L.mvi_compact_plan(C.c_void_p(m.data_ptr()) if P else None, P, C.c_void_p(ws.data_ptr()), nbytes,
                                  C.c_void_p(cnt.data_ptr()), st), "compact_plan")
        n_keep = int(cnt.item())                                # the one host synchronisation: output sizes
        outs = [t.new_empty((n_keep,) + tuple(t.shape[1:])) for t in srcs]
        for i in range(0, len(srcs), 24):
            chunk = [(s, o) for s, o in zip(srcs[i:i + 24], outs[i:i + 24]) if s[0:1].numel() > 0]
            if not chunk or n_keep == 0:
                continue
            tab = (_lib.CompactTensor * len(chunk))()
            for e, (s, o) in zip(tab, chunk):
                e.in_, e.out, e.width = s.data_ptr(), o.data_ptr(), s[0].numel()
            _check(L.mvi_compact_gather(tab, len(chunk), P, n_keep, C.c_void_p(ws.data_ptr()), st), "compact_gather")
    return outs


def prune_optimizer_state(optimizer, keep_mask, extra=()):
    """_prune_optimizer + prune_points (gaussian_model.py:351-382) for a torch.optim.Adam / FusedAdam whose groups hold
    one parameter each: parameters, exp_avg, exp_avg_sq and the `extra` per-Gaussian tensors are compacted together.
    Returns ({group name: new nn.Parameter}, [compacted extras]); optimizer.state is re-keyed like the reference does."""
    groups = [g for g in optimizer.param_groups]
    flat, slots = [], []
    for g in groups:
        p = g["params"][0]
        stt = optimizer.state.get(p, None)
        flat.append(p.data)
        slots.append((g, "param"))
        if stt is not None and "exp_avg" in stt:
            flat += [stt["exp_avg"], stt["exp_avg_sq"]]
            slots += [(g, "exp_avg"), (g, "exp_avg_sq")]
    outs = compact_rows(keep_mask, flat + list(extra))
    new = {}
    it = iter(outs)
    by_group = {}
    for (g, kind) in slots:
        by_group.setdefault(id(g), {})[kind] = next(it)
    for g in groups:
        old = g["params"][0]
        stt = optimizer.state.get(old, None)
        r = by_group[id(g)]
        if stt is not None:
            if "exp_avg" in r:
                stt["exp_avg"], stt["exp_avg_sq"] = r["exp_avg"], r["exp_avg_sq"]
            del optimizer.state[old]
        g["params"][0] = torch.nn.Parameter(r["param"].requires_grad_(True))
        if stt is not None:
            optimizer.state[g["params"][0]] = stt
        new[g.get("name", len(new))] = g["params"][0]
    return new, list(it)


def extend_optimizer_state(optimizer, tensors_dict):
    """cat_tensors_to_optimizer (gs-simp/scene/gaussian_model.py:384-404) for torch.optim.Adam / FusedAdam whose groups hold
    one named parameter each: every parameter gets the new rows of tensors_dict[group name] appended, its Adam moments get
    zero rows, optimizer.state is re-keyed like the reference does. Appending is a device-to-device copy per tensor (nothing
    to fuse): the moments are grown as allocate + copy + memset of the tail (the reference materialises a zeros_like per moment
    and concatenates: two more passes over the new rows and a temporary each). dropin.patch_gs_simp puts it behind the model's
    cat_tensors_to_optimizer. Returns {group name: new nn.Parameter}."""
    new = {}

    def grow(t, n_new):
        out = t.new_empty((t.shape[0] + n_new,) + tuple(t.shape[1:]))
        out[:t.shape[0]].copy_(t)
        out[t.shape[0]:].zero_()
        return out
    for g in optimizer.param_groups:
        assert len(g["params"]) == 1
        old = g["params"][0]
        ext = tensors_dict[g["name"]]
        stt = optimizer.state.get(old, None)
        if stt is not None:
            if "exp_avg" in stt:
                stt["exp_avg"], stt["exp_avg_sq"] = grow(stt["exp_avg"], ext.shape[0]), grow(stt["exp_avg_sq"], ext.shape[0])
            del optimizer.state[old]
        g["params"][0] = torch.nn.Parameter(torch.cat((old.data, ext), dim=0).requires_grad_(True))
        if stt is not None:
            optimizer.state[g["params"][0]] = stt
        new[g["name"]] = g["params"][0]
    return new
