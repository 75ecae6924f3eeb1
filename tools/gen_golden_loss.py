"""Golden vectors for the photometric loss from the IMPORTED reference (run in the build container only):
gs-simp/utils/loss_utils.py l1_loss / ssim, combined as gs-simp/train.py:91-92 and, masked, as
gs-simp/inpaint_rec.py:117-123. Writes tests/golden/loss_small.npz (inputs, loss values, autograd gradients)."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/gs-simp"
sys.path.insert(0, REF)
from utils.loss_utils import l1_loss, ssim          # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "loss_small.npz")
out = {}
for name, (H, W, seed, masked, lam) in {"a": (37, 53, 0, False, 0.2), "b": (16, 16, 1, False, 0.2), "c": (40, 33, 2, True, 0.2),
                                        "d": (9, 70, 3, False, 0.5), "e": (64, 48, 4, True, 0.0)}.items():
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(3, H, W, generator=g)
    img = (gt + 0.25 * torch.randn(3, H, W, generator=g)).clamp(0, 1).requires_grad_(True)
    mask = (torch.rand(1, H, W, generator=g) > 0.6).float() if masked else None
    if masked:                                                      # inpaint_rec.py:120-123
        pd, tg = img * (1.0 - mask), gt * (1.0 - mask)
    else:
        pd, tg = img, gt
    Ll1 = l1_loss(pd, tg)
    s = ssim(pd, tg)
    loss = (1.0 - lam) * Ll1 + lam * (1.0 - s)
    loss.backward()
    out[f"{name}_image"] = img.detach().numpy()
    out[f"{name}_gt"] = gt.numpy()
    if masked:
        out[f"{name}_mask"] = mask.numpy()
    out[f"{name}_lambda"] = np.float32(lam)
    out[f"{name}_loss"] = np.float32(loss.item())
    out[f"{name}_l1"] = np.float32(Ll1.item())
    out[f"{name}_ssim"] = np.float32(s.item())
    out[f"{name}_grad"] = img.grad.numpy()
np.savez_compressed(OUT, **out)
print("wrote", os.path.normpath(OUT), sorted(k for k in out if k.endswith("_loss")))
