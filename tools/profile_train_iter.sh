#!/bin/bash
# Kernel-trace stats of the fused training iteration (run on the GPU box via gpurun). Usage: tools/profile_train_iter.sh <tag>
TAG=${1:-trainprof}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/bt.py <<PY
import sys, json
sys.path.insert(0, "$R")
from multiview_inpaint_amd import bench_train
print(json.dumps(bench_train.run("hip_raw", 10, 2)))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /tmp/bt.py > $OUT/run.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
rm -rf $OUT/trace
python3 $R/tools/stats_table.py $OUT/kernel_stats.csv 12 > $OUT/kernels_per_iteration.txt
cat $OUT/kernels_per_iteration.txt
