"""CPU: the photometric-loss oracle (oracle/loss_oracle.py) against the golden vectors generated from the imported
reference (tests/golden/loss_small.npz, tools/gen_golden_loss.py: gs-simp/utils/loss_utils.py + the train-loop
combination, values and autograd gradients)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import loss_oracle as lo  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "loss_small.npz"))
CASES = sorted({k.split("_")[0] for k in G.files})


def case(name):
    mask = G[f"{name}_mask"] if f"{name}_mask" in G.files else None
    return G[f"{name}_image"], G[f"{name}_gt"], mask, float(G[f"{name}_lambda"])


@pytest.mark.parametrize("name", CASES)
def test_loss_oracle_matches_reference_golden(name):
    img, gt, mask, lam = case(name)
    r = lo.photometric_loss(img, gt, lam, None if mask is None else 1.0 - mask[0])
    assert abs(r["loss"] - float(G[f"{name}_loss"])) < 2e-6 * max(1.0, abs(r["loss"]))
    assert abs(r["l1"] - float(G[f"{name}_l1"])) < 2e-6
    assert abs(r["ssim"] - float(G[f"{name}_ssim"])) < 2e-6
    g_ref = G[f"{name}_grad"].astype(np.float64)
    scale = np.abs(g_ref).max()
    assert np.abs(r["grad"] - g_ref).max() < 2e-5 * scale        # the reference gradient itself is fp32 autograd


def test_window_matches_reference_construction():
    w = lo.window_1d()
    assert w.dtype == np.float32 and abs(float(w.sum()) - 1.0) < 1e-6 and np.argmax(w) == 5
    assert np.allclose(w, w[::-1])
