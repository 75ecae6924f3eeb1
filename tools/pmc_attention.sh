#!/bin/bash
# MFMA utilisation of the attention kernel at the largest SVD shape (B 28, H 5, S 9216, D 64, bf16) from PMC counters,
# separate passes (run on the MI355X box via gpurun): tools/pmc_attention.sh <tag> [variant] -> gpurun_out/<tag>/pmc_attention.txt
# The target is the standalone harness tools/attn_dev/attn_check (C-ABI, no Python): build it first (see its header).
TAG=${1:-attn_pmc}
export MVI_ATTN_VARIANT=${2:-0}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
BIN=$R/tools/attn_dev/attn_check
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- $BIN bench1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/p2 -- $BIN bench1 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/p4 -- $BIN bench1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p3 -- $BIN bench1 > /dev/null 2>&1
{
  echo "attention kernel, B 28 H 5 S 9216 D 64 bf16 (tools/attn_dev/attn_check bench1, MVI_ATTN_VARIANT=$MVI_ATTN_VARIANT), rocprofv3 --pmc, mean per dispatch"
  python3 $R/tools/pmc_summary.py $OUT/p1 attn_flash
  python3 $R/tools/pmc_summary.py $OUT/p2 attn_flash
  python3 $R/tools/pmc_summary.py $OUT/p4 attn_flash
  f=$(find $OUT/p3 -name "*kernel_stats.csv" | head -1)
  grep attn_flash $f | head -2
} > $OUT/pmc_attention.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
cat $OUT/pmc_attention.txt
