#!/bin/bash
# Issue-side PMC passes for the render kernels (run on the MI355X box through gpurun). Usage: tools/pmc_render.sh <tag>
set -u
TAG=${1:-pmcr}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 1 --path raster --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_EXP_GDS --output-format csv -d $OUT/p2 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_IFETCH SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- $B > /dev/null 2>&1
for p in p1 p2 p3; do python3 $R/tools/pmc_summary.py $OUT/$p > $OUT/$p.txt 2>&1; done
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
grep -A 9 "render_backward\|render_forward" $OUT/p1.txt $OUT/p2.txt $OUT/p3.txt
