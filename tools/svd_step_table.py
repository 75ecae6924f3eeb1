"""Per-kernel table of ONE SVD denoise step from `rocprofv3 --kernel-trace --stats` over a run of N identical steps
(bench_svd: warm-up + timed + instrumented passes): calls and time per step = totals / N. Kernels whose call count is not a
multiple of N (weight initialisation, one-off copies) are listed apart. Usage: python tools/svd_step_table.py kernel_stats.csv N"""
import csv
import sys

path, n = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
per_step, once = [], []
for r in rows:
    calls, tot_ns = int(r["Calls"]), float(r["TotalDurationNs"])
    (per_step if calls % n == 0 else once).append((r["Name"].split("(")[0].replace("void ", "")[:110], calls, tot_ns, float(r["AverageNs"])))
per_step.sort(key=lambda t: -t[2])
tot = sum(t[2] for t in per_step) / n / 1e6
print(f"one denoise step = totals of {n} identical steps / {n}: {sum(t[1] for t in per_step) // n} kernels, {tot:.2f} ms of kernel time per step")
print(f"{'kernel':110s} {'calls/step':>10s} {'avg us':>9s} {'ms/step':>9s}")
for name, calls, tot_ns, avg in per_step:
    print(f"{name:110s} {calls // n:10d} {avg / 1e3:9.1f} {tot_ns / n / 1e6:9.3f}")
print(f"\nnot per step (call count not a multiple of {n}): {len(once)} kernels, {sum(t[2] for t in once) / 1e6:.1f} ms in total")
