"""EXPERIMENT ("de-phased epilogues"): the forms of csrc/linear_n320.hip with the odd CUs of the first round of blocks
started late (MVI_N320_DEPHASE, in quarters of the tile's main loop), so that half the chip is in its epilogue while the other half is in
its main loop. One process per setting: MVI_N320_DEPHASE=<n> python tools/experiments/n320_dephase.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from multiview_inpaint_amd.svd import hip_ops  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)


def timed(fn, rounds=5, reps=10):
    ts = []
    for _ in range(rounds):
        fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) / reps)
    return sorted(ts)[len(ts) // 2] * 1e3


out = []
rows = 28 * 9216
x, w, b = rn(rows, 1280).bfloat16(), (rn(320, 1280) * 0.03).bfloat16(), rn(320)
res, lw, lb = rn(rows, 320).bfloat16(), rn(320), rn(320)
out.append(("ln K=1280 level 0", timed(lambda: hip_ops.linear_n320_add_layer_norm(x, w, b, lw, lb, 1e-5, resid=res))))
out.append(("plain K=1280 -> 320 level 0", timed(lambda: hip_ops.linear_n320(x, w, b))))
x3 = rn(rows, 320).bfloat16()
w3 = (rn(320, 320) * 0.05).bfloat16()
out.append(("ln K=320 level 0", timed(lambda: hip_ops.linear_n320_add_layer_norm(x3, w3, b, lw, lb, 1e-5, resid=res))))
del x, x3, res
x1, w1, b1 = rn(28 * 2304, 640).bfloat16(), (rn(640, 640) * 0.04).bfloat16(), rn(640)
out.append(("plain K=640 -> 640 level 1", timed(lambda: hip_ops.linear_n320(x1, w1, b1))))
wg, bg = (rn(2 * 2560, 640) * 0.04).bfloat16(), rn(2 * 2560).bfloat16()
out.append(("geglu K=640 level 1", timed(lambda: hip_ops.ff_geglu_n320(x1, wg, bg))))
x2, w2, b2 = rn(28 * 2304, 2560).bfloat16(), (rn(640, 2560) * 0.02).bfloat16(), rn(640)
out.append(("plain K=2560 -> 640 level 1", timed(lambda: hip_ops.linear_n320(x2, w2, b2))))
tok = rn(28, 9216, 320).bfloat16()
wt = hip_ops.conv3x3_n320_weight((rn(320, 320, 3, 3) * 0.02).bfloat16())
out.append(("conv3x3 level 0 320 -> 320", timed(lambda: hip_ops.conv3x3_n320(tok, wt, None, 72, 128))))
wt3 = hip_ops.conv3t_n320_weight((rn(320, 320, 3, 1, 1) * 0.03).bfloat16())
out.append(("conv3t level 0 320", timed(lambda: hip_ops.conv3t_n320(tok, wt3, None, 14))))
print(f"dephase={os.environ.get('MVI_N320_DEPHASE', '0')}: " + "  ".join(f"{n} {t:.1f}" for n, t in out), flush=True)
