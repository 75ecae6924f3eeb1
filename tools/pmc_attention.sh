#!/bin/bash
# MFMA utilisation of the attention kernel at the largest SVD shape (B 28, H 5, S 9216, D 64, bf16) from PMC counters,
# separate passes (run on the MI355X box via gpurun): tools/pmc_attention.sh <tag> -> gpurun_out/<tag>/pmc_attention.txt
TAG=${1:-attn_pmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 $R/tools/attn_one.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/p2 -- python3 $R/tools/attn_one.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p3 -- python3 $R/tools/attn_one.py > /dev/null 2>&1
{
  echo "attention kernel, B 28 H 5 S 9216 D 64 bf16 (tools/attn_one.py), rocprofv3 --pmc, mean per dispatch"
  python3 $R/tools/pmc_summary.py $OUT/p1 attn_flash
  python3 $R/tools/pmc_summary.py $OUT/p2 attn_flash
  f=$(find $OUT/p3 -name "*kernel_stats.csv" | head -1)
  grep attn_flash $f | head -2
} > $OUT/pmc_attention.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
cat $OUT/pmc_attention.txt
