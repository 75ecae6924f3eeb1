#!/usr/bin/env python3
"""Where does the reduced-precision error of the production-width networks come from?

tests/test_unet_ops_gpu.py::test_production_width_nets_run_the_round3_kernels_within_the_reference_autocast_budget holds the HIP path
in bf16 / f16 to a multiple of the REFERENCE's own autocast error (tests/golden/sgm_c320.npz). In f16 the observed multiple was 1.56:
this tool re-runs the same three networks with ONE kernel family at a time handed back to the library / to the other form and
prints the multiple per recorded tensor, so the op that carries the excess can be named.

    python tools/diag_c320_precision.py [--dtype f16|bf16]        (parent: spawns one child per variant BEFORE touching the GPU)

Variants (module switches of multiview_inpaint_amd/svd, set in the child before the first launch):
    default            what ships
    attn_exact         MVI_ATTN_FOLD_SCALE=0: softmax scale applied to the fp32 scores (f16 folds it into Q by default)
    no_conv_n320       3x3 / (3,1,1) convolutions by the library instead of csrc/linear_n320.hip
    no_time_tokens     temporal ResBlock on the NCHW frames path
    no_nhwc            ResBlocks on the NCHW path (library convolutions, NCHW GroupNorm kernels)
    no_k320            level-0 projections / GEGLU by the library (csrc/ff_geglu.hip, linear_n320 plain off)
    no_temporal_mfma   temporal attention by the fp32-math kernel
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {
    "default": {},
    "attn_exact": {"MVI_ATTN_FOLD_SCALE": "0"},
    "attn_fold": {"MVI_ATTN_FOLD_SCALE": "1"},
    "no_conv_n320": {"MVI_SVD_CONV_N320": "0"},
    "no_time_tokens": {"MVI_SVD_TIME_STACK_TOKENS": "0"},
    "no_nhwc": {"MVI_SVD_NHWC_CONVS": "0"},
    "no_k320": {"MVI_K320": "0", "MVI_N320": "0"},
    "no_temporal_mfma": {"MVI_ATTN_TEMPORAL_MFMA": "0"},
}


def child(dtype_name):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "multiview_inpaint_amd", "dropin"))
    import numpy as np
    import torch
    import svd_helpers as H
    from sgm.modules.diffusionmodules.video_model import VideoUNet
    from models.csvd import ControlNet, ControlledVideoUNet
    from multiview_inpaint_amd.svd import layers as LY
    dtype = torch.float16 if dtype_name == "f16" else torch.bfloat16
    tag = "f16ac" if dtype_name == "f16" else "bf16ac"
    G = np.load(os.path.join(ROOT, "tests", "golden", "sgm_c320.npz"))
    unet = VideoUNet(**H.SMALL_UNET320).eval()
    unet.load_state_dict(H.seeded_state_dict(unet, 51))
    cunet = ControlledVideoUNet(**H.SMALL_UNET320).eval()
    cunet.load_state_dict(H.seeded_state_dict(cunet, 51))
    cnet = ControlNet(**H.SMALL_CTRL320).eval()
    cnet.load_state_dict(H.seeded_state_dict(cnet, 52))
    unet, cunet, cnet = (m.cuda().to(dtype) for m in (unet, cunet, cnet))
    inp = H.seeded_inputs(53, hw=H.LATENT_HW320, cfg=H.SMALL_UNET320)
    inp["image_only_indicator"][0, 1] = 1.0
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    kw = dict(num_video_frames=H.T_FRAMES, image_only_indicator=inp["image_only_indicator"])
    xin = torch.cat([inp["x"], inp["concat"]], 1).to(dtype)
    tt = 0.25 * inp["sigma"].log()
    ctx, vec, hint = inp["crossattn"].to(dtype), inp["vector"].to(dtype), inp["control_hint"].to(dtype)
    LY.CONV_N320_MIN_BLOCKS = 1
    probes = {}
    for name in H.C320_PROBES:
        unet.get_submodule(name).register_forward_hook(
            lambda m, i, o, name=name: probes.__setitem__(name, o.detach().float()[:, ::4, ::2, ::2].contiguous()))
    with torch.no_grad():
        y = unet(xin, tt, ctx, vec, **kw)
        ctrls = cnet(xin, hint, tt, ctx, vec, **kw)
        yc = cunet(xin, tt, ctx, vec, control=list(ctrls), **kw)
    torch.cuda.synchronize()

    def err(a, b):
        a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
        d = (a - b).abs()
        return float(d.max() / (b.abs().max() + 1e-12)), float(d.pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12))

    out = {}
    for name, got in [("unet_out", y), ("cunet_out", yc), ("ctrl_last", ctrls[-1])] + [("probe_" + k, probes[k]) for k in H.C320_PROBES]:
        ref = G[name + "_f32"]
        e_max, e_rms = err(got.float(), ref)
        r_max, r_rms = err(torch.tensor(G[name + "_" + tag]), ref)
        out[name] = [round(e_max / r_max, 3), round(e_rms / r_rms, 3)]
    print("RESULT " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--variants", default=",".join(VARIANTS))
    a = ap.parse_args()
    if a.child:
        return child(a.dtype)
    rows = {}
    for v in a.variants.split(","):
        env = dict(os.environ, MVI_STRICT="0", **VARIANTS[v])
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--dtype", a.dtype], env=env, cwd=ROOT,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        line = next((ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")), None)
        if line is None:
            print(f"{v}: FAILED rc={p.returncode}\n" + "\n".join(p.stderr.strip().splitlines()[-8:]), flush=True)
            continue
        rows[v] = json.loads(line[7:])
        worst = max(max(x) for x in rows[v].values())
        print(f"{a.dtype} {v:18s} worst multiple of the reference's own autocast error {worst:.2f}  " +
              "  ".join(f"{k}={x[0]:.2f}/{x[1]:.2f}" for k, x in rows[v].items()), flush=True)
    print("JSON " + json.dumps({"dtype": a.dtype, "multiples_max_rms": rows}))


if __name__ == "__main__":
    main()
