"""Differentiable dense restatement of the rasterizer in torch (fp64) — used ONLY to check the
analytic backward of oracle/raster_oracle.c (and through it the HIP backward) with
torch.autograd on scenes of <=~100 Gaussians and <=48x48 pixels.

TEST INFRASTRUCTURE ONLY. PARITY UNPINNED against the third-party CUDA extension (see
oracle/raster_oracle.c header). Written from the math (EWA splatting + front-to-back
compositing), vectorised over [pixels, gaussians]; shares no code with the C oracle.

Deliberate non-smooth conventions kept from the restated algorithm so autograd reproduces the
same gradient the extension returns:
  * the frustum clamp of x/z, y/z in the Jacobian passes no gradient (clamped value detached);
  * alpha = min(0.99, o*G) is straight-through (gradient as if unclamped);
  * radii / tile rectangles / the median depth carry no gradient;
  * means2D receives d(loss)/d(ndc offset): pixel-space gradient times W/2, H/2
    (consumer: gs-simp/scene/gaussian_model.py:482-484).
"""

import torch

TILE = 16
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def _sh_color(deg, sh, d):
    """sh [P,M,3], d [P,3] unit -> [P,3] (gs-simp/utils/sh_utils.py:57-112)"""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = SH_C0 * sh[:, 0]
    if deg > 0:
        r = r - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
             + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        r = (r + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
             + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
             + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
             + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r


def _rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)
    return R.view(-1, 3, 3)


def render(cam, bg, sh_degree, means3D, means2D, opacities, shs=None, colors_precomp=None,
           scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0):
    """All tensor arguments float64 torch tensors (leaf tensors may require grad).
    Returns color [3,H,W], depth [1,H,W], radii [P], aux dict."""
    dt = torch.float64
    W, H = cam["W"], cam["H"]
    V = torch.as_tensor(cam["viewmatrix"], dtype=dt)      # row-vector convention: p_view = [p,1] @ V
    PM = torch.as_tensor(cam["projmatrix"], dtype=dt)
    campos = torch.as_tensor(cam["campos"], dtype=dt)
    bg = torch.as_tensor(bg, dtype=dt)
    tfx, tfy = cam["tanfovx"], cam["tanfovy"]
    fx, fy = W / (2 * tfx), H / (2 * tfy)
    P = means3D.shape[0]
    ph = torch.cat([means3D, torch.ones(P, 1, dtype=dt)], 1)
    t = ph @ V[:, :3]
    hom = ph @ PM
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None] + means2D[:, :2]
    pix = torch.stack([((ndc[:, 0] + 1) * W - 1) * 0.5, ((ndc[:, 1] + 1) * H - 1) * 0.5], 1)

    if cov3D_precomp is not None:
        c6 = cov3D_precomp
        Sig = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4],
                           c6[:, 2], c6[:, 4], c6[:, 5]], -1).view(-1, 3, 3)
    else:
        Mx = _rot(rotations) * (scale_modifier * scales)[:, None, :]
        Sig = Mx @ Mx.transpose(1, 2)

    tz = t[:, 2]
    limx, limy = 1.3 * tfx, 1.3 * tfy
    rx, ry = t[:, 0] / tz, t[:, 1] / tz
    tx = torch.where((rx < -limx) | (rx > limx), (rx.clamp(-limx, limx) * tz).detach(), t[:, 0])
    ty = torch.where((ry < -limy) | (ry > limy), (ry.clamp(-limy, limy) * tz).detach(), t[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz)], -1).view(-1, 2, 3)
    Rw = V[:3, :3].T                                       # world->view rotation (column-vector form)
    Tm = J @ Rw
    cov = Tm @ Sig @ Tm.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    conA, conB, conC = c / det, -b / det, a / det
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3 * torch.sqrt(lam)).to(torch.int64)

    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    pd, rf = pix.detach(), radius.to(dt)

    def _t(v, g):
        return torch.clamp(torch.trunc(v / TILE).to(torch.int64), 0, g)
    x0, y0 = _t(pd[:, 0] - rf, gx), _t(pd[:, 1] - rf, gy)
    x1, y1 = _t(pd[:, 0] + rf + TILE - 1, gx), _t(pd[:, 1] + rf + TILE - 1, gy)
    visible = (tz.detach() > 0.2) & (det.detach() != 0) & ((x1 - x0) * (y1 - y0) > 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius))

    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(_sh_color(sh_degree, shs, d) + 0.5, 0.0)

    # depth order, stable by index (== sort by (tile, depth-bits) restricted to one tile)
    order = torch.sort(tz.detach().to(torch.float32), stable=True).indices
    order = order[visible[order]]
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pxs, pys = xs.reshape(-1), ys.reshape(-1)
    txs, tys = pxs // TILE, pys // TILE
    in_rect = ((txs[:, None] >= x0[order][None]) & (txs[:, None] < x1[order][None]) &
               (tys[:, None] >= y0[order][None]) & (tys[:, None] < y1[order][None]))
    dx = pix[order, 0][None] - pxs[:, None].to(dt)
    dy = pix[order, 1][None] - pys[:, None].to(dt)
    power = -0.5 * (conA[order][None] * dx * dx + conC[order][None] * dy * dy) - conB[order][None] * dx * dy
    G = torch.exp(torch.clamp_max(power, 0.0))
    alpha_raw = opacities.reshape(-1)[order][None] * G
    alpha = alpha_raw + (alpha_raw.clamp(max=0.99) - alpha_raw).detach()
    live = in_rect & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
    a_eff = torch.where(live, alpha, torch.zeros_like(alpha))
    # sequential termination rule: stop at the first Gaussian with T*(1-alpha) < 1e-4
    Tprev = torch.cumprod(torch.cat([torch.ones(a_eff.shape[0], 1, dtype=dt), 1 - a_eff], 1), 1)
    test_T = Tprev[:, 1:].detach()
    stopped = torch.cumsum((live & (test_T < 1e-4)).to(torch.int64), 1) > 0
    a_eff = torch.where(stopped, torch.zeros_like(a_eff), a_eff)
    Tprev = torch.cumprod(torch.cat([torch.ones(a_eff.shape[0], 1, dtype=dt), 1 - a_eff], 1), 1)
    w = a_eff * Tprev[:, :-1]
    T_final = Tprev[:, -1]
    color = w @ rgb[order] + T_final[:, None] * bg[None]
    contrib = live & ~stopped
    Tb, Ta = Tprev[:, :-1].detach(), Tprev[:, 1:].detach()
    med = contrib & (Tb > 0.5) & (Ta < 0.5)
    depth = torch.full((H * W,), 15.0, dtype=dt)
    has = med.any(1)
    first = med.to(torch.int64).argmax(1)
    depth[has] = tz.detach()[order][first[has]].to(torch.float32).to(dt)
    idx = torch.cumsum(in_rect.to(torch.int64), 1)      # 1-based position in the tile's own list
    n_contrib = torch.where(contrib, idx, torch.zeros_like(idx)).max(1).values if order.numel() else torch.zeros(H * W, dtype=torch.int64)
    aux = dict(pix=pix, conic=torch.stack([conA, conB, conC], 1), rgb=rgb, tz=tz, order=order,
               final_T=T_final.view(H, W), n_contrib=n_contrib.view(H, W), visible=visible)
    return color.T.reshape(3, H, W), depth.view(1, H, W), radii, aux
