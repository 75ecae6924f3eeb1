"""The block tails of the token-major residual stream (csrc/groupnorm_tokens.hip gt_fused_kernel: mvi_rows_fused_gnstats) at the shapes
of the 14 x 576x1024 step, with and without the next norm's statistics, against the plain three-operand row kernel (add_lerp) as the
streaming yardstick: time and algorithmic GB/s, forms interleaved in one process. Run on the GPU box: python tools/bench_rows_fused.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multiview_inpaint_amd.svd import hip_ops  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)


def timed(fns, rounds=5, reps=10):
    ts = [[] for _ in fns]
    for _ in range(rounds):
        for k, fn in enumerate(fns):
            fn()
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            e.record()
            torch.cuda.synchronize()
            ts[k].append(a.elapsed_time(e) / reps)
    return [sorted(t)[len(t) // 2] for t in ts]


for (N, S, C) in [(28, 9216, 320), (28, 2304, 640), (28, 576, 1280), (28, 144, 1280)]:
    mk = lambda c=C: torch.randn(N, S, c, device=dev, generator=g).bfloat16()
    a, b, base = mk(), mk(), mk()
    bias = torch.randn(C, device=dev, generator=g)
    alpha = torch.rand(N, device=dev, generator=g)
    w, bb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    nb = a.numel() * 2
    out, st = hip_ops.rows_fused(a, b, bias=bias, groups=32)
    fns = [lambda: hip_ops.rows_fused(a, b, bias=bias, groups=32), lambda: hip_ops.rows_fused(a, b, bias=bias),
           lambda: hip_ops.rows_fused(a, bias=bias, base=base, alpha=alpha, groups=32),
           lambda: hip_ops.rows_fused(a, b, base=base, concat=True, groups=32),
           lambda: hip_ops.add_lerp(a, b, base, alpha.bfloat16()),
           lambda: hip_ops.group_norm_silu_tok2tok(out, 32, w, bb, 1e-5, True, partials=st),
           lambda: hip_ops.group_norm_silu_tok2tok(out, 32, w, bb, 1e-5, True)]
    names = ["add+stats", "add", "blend+stats", "concat+stats", "add_lerp", "norm(pre)", "norm(3 launches)"]
    byts = [3 * nb, 3 * nb, 3 * nb, 5 * nb, 4 * nb, 2 * nb, 3 * nb]
    ms = timed(fns)
    print(f"[{N}, {S}, {C}] bf16: " + "  ".join(f"{n} {t * 1e3:.1f} us ({by / t * 1e-9:.2f} TB/s)" for n, t, by in zip(names, ms, byts)), flush=True)
