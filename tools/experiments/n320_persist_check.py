"""The persistent form of csrc/linear_n320.hip (kPersist: one block per CU walks the tiles, the next tile's first loads under the last chunks
of the one in hand) against the plain grid: outputs bit for bit (same products, same order), then time, forms alternating in one process
(MVI_N320_PERSIST is read per launch). GPU box: timeout -k 10 300 python tools/experiments/n320_persist_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from multiview_inpaint_amd.svd import hip_ops  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)


def both(fn):
    os.environ["MVI_N320_PERSIST"] = "0"
    a = fn()
    os.environ["MVI_N320_PERSIST"] = "1"
    b = fn()
    torch.cuda.synchronize()
    return a, b


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3


cases = []
for dtype in (torch.bfloat16, torch.float16):
    for (rows, K, N, with_bias) in [(28 * 9216, 1280, 320, True), (28 * 2304, 640, 640, True), (28 * 2304, 2560, 640, False), (28 * 2304, 640, 1920, True),
                                    (28 * 576, 1280, 1280, True), (28 * 576 - 100, 5120, 1280, True), (28 * 9216 - 7, 256, 320, True)]:
        x, w = rn(rows, K).to(dtype), (rn(N, K) * K ** -0.5).to(dtype)
        b = rn(N) if with_bias else None
        cases.append((f"linear {rows} x {K} -> {N} {str(dtype)[6:]}", lambda x=x, w=w, b=b: hip_ops.linear_n320(x, w, b)))
    for (rows, K, inner) in [(28 * 2304, 640, 2560), (28 * 576, 1280, 5120), (28 * 2304 - 33, 640, 2560)]:
        x, w, b = rn(rows, K).to(dtype), (rn(2 * inner, K) * K ** -0.5).to(dtype), rn(2 * inner).to(dtype)
        cases.append((f"geglu {rows} x {K} -> 2 x {inner} {str(dtype)[6:]}", lambda x=x, w=w, b=b: hip_ops.ff_geglu_n320(x, w, b)))
bad = 0
for name, fn in cases:
    a, b = both(fn)
    same = torch.equal(a, b)
    bad += not same
    ts = {"0": [], "1": []}
    for _ in range(5):
        for mode in ("0", "1"):
            os.environ["MVI_N320_PERSIST"] = mode
            ts[mode].append(timed(fn))
    t0, t1 = sorted(ts["0"])[2], sorted(ts["1"])[2]
    print(f"{name}: {'identical' if same else 'DIFFERENT (max |diff| %.3e)' % float((a.float() - b.float()).abs().max())}; plain grid {t0:.1f} us, persistent {t1:.1f} us ({(t1 / t0 - 1) * 100:+.1f} %)", flush=True)
print("ALL IDENTICAL" if not bad else f"{bad} CASES DIFFER")
sys.exit(1 if bad else 0)
