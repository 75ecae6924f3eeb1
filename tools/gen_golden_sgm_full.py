"""Generates tests/golden/sgm_full.npz: ONE denoise-network evaluation of the reference at the FULL size of BASELINE.json
configs[3] — ControlNet + ControlledVideoUNet of configs/test/svd_f_est_ctrl_simp1.yaml (1.52 B + 0.68 B parameters), 14 frames on
the 72 x 128 latent, CFG batch 28 — on the CPU: the fp32 output and four subsampled intermediate block outputs, and — as the error budget — the
error of the same evaluation under the reference's own reduced-precision recipe (bf16 autocast over fp32 weights,
models/csvd.py:27-31) against them (two numbers per tensor: max norm and rms). Weights and inputs are seeded (tests/svd_helpers.py: variance-preserving N(0, 1 / fan_in), norm scales 1 + 0.1 N): the GPU
test regenerates them bit for bit. Run ONLY in the build container (~25 GB of memory, tens of minutes on 8 cores).

Usage: python tools/gen_golden_sgm_full.py            (fp32 tensors + bf16-autocast budgets)
       python tools/gen_golden_sgm_full.py --add-f16  (adds f16-autocast budgets to the existing fixture)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, HERE)
import svd_helpers as H  # noqa: E402
from ref_import import import_reference  # noqa: E402

t0 = time.time()
log = lambda *a: print(f"[{time.time() - t0:7.1f} s]", *a, flush=True)
ref = import_reference()
torch.manual_seed(0)
T = H.FULL_T
out = {}

cunet = ref["ControlledVideoUNet"](**H.FULL_UNET).eval()
cunet.load_state_dict(H.seeded_state_dict(cunet, 71), strict=True)
log("ControlledVideoUNet built,", sum(p.numel() for p in cunet.parameters()), "parameters")
cnet = ref["ControlNet"](**H.FULL_CTRL).eval()
cnet.load_state_dict(H.seeded_state_dict(cnet, 72), strict=True)
log("ControlNet built,", sum(p.numel() for p in cnet.parameters()), "parameters")

inp = H.seeded_inputs(73, T=T, hw=H.FULL_HW, cfg=H.FULL_UNET)
inp["image_only_indicator"][0, 1] = 1.0              # one frame blended as an image: the AlphaBlender's other branch
kw = dict(num_video_frames=T, image_only_indicator=inp["image_only_indicator"])
xin = torch.cat([inp["x"], inp["concat"]], 1)
tt = 0.25 * inp["sigma"].log()

_seen = {}
for name in H.FULL_PROBES:
    cunet.get_submodule(name).register_forward_hook(
        lambda m, i, o, name=name: _seen.__setitem__(name, o.detach().float()[H.FULL_SUB].contiguous().numpy()))


def run(tag):
    _seen.clear()
    ctrls = cnet(xin, inp["control_hint"], tt, inp["crossattn"], inp["vector"], **kw)
    log(tag, "ControlNet done,", len(ctrls), "residuals")
    yc = cunet(xin, tt, inp["crossattn"], inp["vector"], control=[c.clone() for c in ctrls], **kw)
    log(tag, "ControlledVideoUNet done")
    out["cunet_out_" + tag] = yc.float().numpy()
    out["ctrl_last_" + tag] = ctrls[-1].float()[H.FULL_SUB].contiguous().numpy()
    for name in H.FULL_PROBES:
        out[f"probe_{name}_{tag}"] = _seen[name]
    return len(ctrls)


path = os.path.join(HERE, "..", "tests", "golden", "sgm_full.npz")
if "--add-f16" in sys.argv:
    # second budget, added to an existing fixture: the same evaluation under f16 autocast (the reference's own GPU recipe,
    # configs/test/svd_f_est_ctrl_simp1.yaml:214) against the fp32 tensors already recorded
    G = dict(np.load(path))
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
        run("f16ac")
    for k in [k for k in G if k.endswith("_f32")]:
        a, b = out[k[:-4] + "_f16ac"].astype(np.float64), G[k].astype(np.float64)
        G["budget_f16_" + k[:-4]] = np.array([np.abs(a - b).max() / np.abs(b).max(), np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean())])
    np.savez_compressed(path, **G)
    log("added f16 budgets:", {k: G[k].tolist() for k in G if k.startswith("budget_f16_")})
    sys.exit(0)

with torch.no_grad():
    n = run("f32")
    with torch.autocast("cpu", dtype=torch.bfloat16):
        run("bf16ac")
out["n_ctrl"] = np.array(n)


def rel_err(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    d = a - b
    return float(np.abs(d).max() / np.abs(b).max()), float(np.sqrt((d ** 2).mean()) / np.sqrt((b ** 2).mean()))


# the fixture keeps the fp32 tensors and, of the autocast run, only what the test needs: its error against them (max norm, rms)
keep = {"n_ctrl": out["n_ctrl"]}
for k in [k for k in out if k.endswith("_f32")]:
    keep[k] = out[k]
    keep["budget_" + k[:-4]] = np.array(rel_err(out[k[:-4] + "_bf16ac"], out[k]), np.float64)
np.savez_compressed(path, **keep)
log("wrote", os.path.normpath(path), f"{os.path.getsize(path) / 1e6:.2f} MB; autocast-vs-fp32 (max, rms): cunet", keep["budget_cunet_out"],
    "mean|cunet_out|", float(np.abs(out["cunet_out_f32"]).mean()), "finite", bool(np.isfinite(out["cunet_out_f32"]).all()))
