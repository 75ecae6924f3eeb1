"""TEST INFRASTRUCTURE — CPU restatement (numpy, fp64 accumulation) of the reference's photometric loss and of its
gradient with respect to the rendered image. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this; the product path is the HIP kernel (multiview_inpaint_amd/csrc/photometric_loss.hip).

Follows gs-simp/utils/loss_utils.py: l1_loss :17-18, gaussian :23-25 (window built in fp32), create_window :27-31
(outer product of the 1-D window, fp32), _ssim :43-62 (depthwise 11x11 conv, zero padding 5, C1 = 0.01^2,
C2 = 0.03^2, mean over all elements), combined as at gs-simp/train.py:91-92 / gs-simp/inpaint_rec.py:117-123:
    loss = (1 - lambda) * l1(x, y) + lambda * (1 - ssim(x, y)),  x = image * w, y = gt * w  (w = 1 - gt_mask or 1).
Pinned against the imported reference by tests/golden/loss_small.npz (tools/gen_golden_loss.py)."""
import numpy as np


def window_1d():
    g = np.array([np.exp(-(x - 11 // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)]).astype(np.float32)
    return (g / g.sum(dtype=np.float32)).astype(np.float32)


def window_2d():
    w = window_1d()
    return (w[:, None] * w[None, :]).astype(np.float32)          # fp32 outer product, as _1D_window.mm(_1D_window.t())


def _conv(img, w2):
    """Depthwise 11x11 correlation with zero padding 5: img [C,H,W] -> [C,H,W] (fp64)."""
    C, H, W = img.shape
    p = np.zeros((C, H + 10, W + 10))
    p[:, 5:5 + H, 5:5 + W] = img
    out = np.zeros((C, H, W))
    for i in range(11):
        for j in range(11):
            out += float(w2[i, j]) * p[:, i:i + H, j:j + W]
    return out


def photometric_loss(image, gt, lambda_dssim=0.2, weight=None, need_grad=True):
    """image, gt [3,H,W]; weight [H,W] or None. Returns dict(loss, l1, ssim, grad) with grad = d loss / d image."""
    image = np.asarray(image, np.float64)
    gt = np.asarray(gt, np.float64)
    wmap = np.ones(image.shape[1:]) if weight is None else np.asarray(weight, np.float64)
    x, y = image * wmap, gt * wmap
    w2 = window_2d()
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2 = _conv(x, w2), _conv(y, w2)
    e11, e22, e12 = _conv(x * x, w2), _conv(y * y, w2), _conv(x * y, w2)
    s1, s2, s12 = e11 - mu1 * mu1, e22 - mu2 * mu2, e12 - mu1 * mu2
    A1, A2 = 2 * mu1 * mu2 + C1, 2 * s12 + C2
    B1, B2 = mu1 * mu1 + mu2 * mu2 + C1, s1 + s2 + C2
    smap = A1 * A2 / (B1 * B2)
    n = float(x.size)
    l1, ssim = np.abs(x - y).sum() / n, smap.sum() / n
    out = dict(loss=(1 - lambda_dssim) * l1 + lambda_dssim * (1 - ssim), l1=l1, ssim=ssim, grad=None)
    if need_grad:
        d_mu1 = (2 * mu2 * (A2 - A1) - 2 * mu1 * smap * (B2 - B1)) / (B1 * B2)
        d_e11 = -smap / B2
        d_e12 = 2 * A1 / (B1 * B2)
        # the window is symmetric and the padding zero: the adjoint of the correlation is the same correlation
        g = _conv(d_mu1, w2) + 2 * x * _conv(d_e11, w2) + y * _conv(d_e12, w2)
        gx = (1 - lambda_dssim) * np.sign(x - y) / n - lambda_dssim * g / n
        out["grad"] = gx * wmap
    return out


def knn3_mean_dist2(points):
    """distCUDA2 of simple_knn (call sites gs-simp/scene/gaussian_model.py:134, :546, :623; the package itself is a
    third-party CUDA plug-in absent from the tree — PARITY UNPINNED beyond its published behaviour): mean of the squared
    distances to the 3 nearest other points, exact, by brute force in fp64 chunks."""
    p = np.asarray(points, np.float64)
    n = p.shape[0]
    out = np.empty(n)
    for a in range(0, n, 512):
        d = ((p[a:a + 512, None, :] - p[None, :, :]) ** 2).sum(-1)
        d[np.arange(d.shape[0]), np.arange(a, a + d.shape[0])] = np.inf
        out[a:a + 512] = np.sort(d, axis=1)[:, :3].mean(1)
    return out
