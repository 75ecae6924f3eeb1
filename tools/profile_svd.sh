#!/bin/bash
# Kernel-trace stats of the full-size SVD denoise step (run on the GPU box via gpurun).
TAG=${1:-svdprof}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/svd_step.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from multiview_inpaint_amd.svd import bench_svd
r = bench_svd.run_gpu(torch.device("cuda"), steps=2, warmup=1, sample_steps=0)
print(r)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /tmp/svd_step.py > $OUT/run.log 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - <<PY
# steady state only: the last timed step = everything after the second-to-last occurrence of the step's first kernel
import csv, collections
rows = [r for r in csv.DictReader(open("$f"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step contains exactly 23 flash-attention launches; cut at the 23rd-from-last one's step start
fl = [i for i, r in enumerate(rows) if "attn_flash" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]]
start_i = fl[-23]
# walk back to the beginning of that step: the hint stem's first conv comes ~60 kernels earlier; use a time gap instead
t_first = int(rows[start_i]["Start_Timestamp"])
step_ms = (int(rows[-1]["End_Timestamp"]) - int(rows[fl[-46]]["Start_Timestamp"])) / 2e6 if len(fl) >= 46 else None
last = [r for r in rows if int(r["Start_Timestamp"]) >= int(rows[fl[-23]]["Start_Timestamp"]) - 30_000_000]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in last:
    k = r["Kernel_Name"]
    k = (k[:60] + " ... " + k[k.find("native::", 80):k.find("native::", 80) + 120]) if "elementwise_kernel" in k or k.startswith("void at::native::") else k.split("(")[0][:100]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(v[1] for v in agg.values())
with open("$OUT/steady_step_kernels.txt", "w") as fo:
    print(f"steady-state step: {len(last)} kernels, {tot:.1f} ms of kernel time (window ~ last step)", file=fo)
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{k:190s} {n:5d} {ms:9.2f} ms {100 * ms / tot:5.1f}%", file=fo)
print(open("$OUT/steady_step_kernels.txt").read())
PY
rm -rf $OUT/trace
tail -2 $OUT/run.log
