"""Host side of include/mvi_train_ops.h: the ops either side of the rasterizer inside the timed region of the 3DGS
training loop, with the reference's own function names so that a training script changes one import
line (`from utils.loss_utils import l1_loss, ssim` -> `from multiview_inpaint_amd.train_ops import l1_loss, ssim`;
gs-simp/train.py:17, gs-simp/inpaint_rec.py:16).

GPU tensors run the HIP kernels through the C-ABI and raise if the library is missing; there is no CPU path here
(the CPU restatement lives in oracle/loss_oracle.py and is test infrastructure)."""
import ctypes as C

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


def l1_loss(network_output, gt):
    """gs-simp/utils/loss_utils.py:17-18 (same name and signature)."""
    return _PhotometricLoss.apply(network_output, gt, None, 0.0, 1)


def ssim(img1, img2, window_size=11, size_average=True):
    """gs-simp/utils/loss_utils.py:33-41 (same name and signature; the window size used by every caller is 11)."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("ssim: only window_size=11, size_average=True (the only form the training scripts use)")
    return _PhotometricLoss.apply(img1, img2, None, 1.0, 2)
