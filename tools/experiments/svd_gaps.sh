#!/bin/bash
# GPU idle time inside a steady-state SVD denoise step (kernel trace; run on the GPU box).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/svd_gaps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/svd_step.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from multiview_inpaint_amd.svd import bench_svd
r = bench_svd.run_gpu(torch.device("cuda"), steps=2, warmup=1, sample_steps=0)
print(r["ms_per_step"])
PY
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 /tmp/svd_step.py > $OUT/run.log 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows = sorted(csv.DictReader(open("$f")), key=lambda r: int(r["Start_Timestamp"]))
fl = [i for i, r in enumerate(rows) if "attn_flash_kernel" in r["Kernel_Name"]]
# steps: warm-up, 2 timed, 2 instrumented = 5 x 23 launches; take the second timed step (3rd block of 23)
a, b = fl[2 * 23], fl[3 * 23]
seg = rows[a:b]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
gaps = sorted(((int(q["Start_Timestamp"]) - int(p["End_Timestamp"])) for p, q in zip(seg, seg[1:])), reverse=True)
print(f"one timed step: {len(seg)} kernels, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms, idle {100*(span-busy)/span:.1f} %")
print("largest gaps (us):", [round(g / 1e3, 1) for g in gaps[:12]], " gaps > 5 us:", sum(g > 5000 for g in gaps), " total of those (ms):", round(sum(g for g in gaps if g > 5000) / 1e6, 2))
PY
rm -rf $OUT/trace
tail -1 $OUT/run.log
