#!/bin/bash
# One-step kernel table of the SVD denoise step for the SHIPPED build (profiles/*_svd_one_step_kernels.txt): rocprofv3 kernel stats
# over bench_svd's 12 identical steps (2 warm-up + 5 timed + 5 instrumented; no sample loop: it caches the hint stem), divided by 12. Usage (gpurun): tools/profile_svd_step_table.sh <tag>
TAG=${1:-svdtab}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R && export TMPDIR=/tmp
# one stream: with the ControlNet on a side stream (the engine's default since round 6) a kernel's duration includes what ran beside it
export MVI_SVD_TWO_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 -m multiview_inpaint_amd.svd.bench_svd --steps 5 --warmup 2 --sample-steps 0 --weights bf16 > $OUT/bench_under_trace.json 2> $OUT/err.log
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv; rm -rf $OUT/trace
N=$(python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
c=[int(r["Calls"]) for r in rows if "attn_flash8" in r["Name"]]
print(sum(c)//14 if c else 12)
PY
)
python3 tools/svd_step_table.py $OUT/kernel_stats.csv $N > $OUT/svd_one_step_kernels.txt
head -40 $OUT/svd_one_step_kernels.txt; tail -1 $OUT/bench_under_trace.json | cut -c1-300
