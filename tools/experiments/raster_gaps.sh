#!/bin/bash
# GPU idle gaps between consecutive kernels of steady-state rasterizer steps (run on the GPU box).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/gaps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 6 --warmup 2 --path raster --no-cpu-baseline > $OUT/bench.json 2>/dev/null
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows = sorted(csv.DictReader(open("$f")), key=lambda r: int(r["Start_Timestamp"]))
# steady state: from the 4th-from-last preprocess_forward launch to the end
idx = [i for i, r in enumerate(rows) if "preprocess_forward" in r["Kernel_Name"]]
a, b = idx[-4], idx[-1]
seg = rows[a:b]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
print(f"3 steps: span {span/3e3:.1f} us/step, kernels busy {busy/3e3:.1f} us/step, idle {100*(span-busy)/span:.1f}%")
gaps = []
for p, q in zip(seg, seg[1:] + [rows[b]]):
    g = int(q["Start_Timestamp"]) - int(p["End_Timestamp"])
    gaps.append((g, p["Kernel_Name"].split("(")[0][-40:], q["Kernel_Name"].split("(")[0][-40:]))
import collections
agg = collections.defaultdict(lambda: [0, 0])
for g, p, q in gaps:
    agg[(p, q)][0] += g; agg[(p, q)][1] += 1
for (p, q), (g, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{g/3e3:7.1f} us/step  x{n/3:.0f}  {p:40s} -> {q}")
PY
rm -rf $OUT/trace
