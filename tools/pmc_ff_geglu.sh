#!/bin/bash
# Matrix-pipe / VALU / LDS counters of the K = 320 MFMA kernels (csrc/ff_geglu.hip) at the level-0 shapes, separate PMC passes
# (run on the MI355X box via gpurun): tools/pmc_ff_geglu.sh <tag> -> gpurun_out/<tag>/pmc_ff_geglu.txt
TAG=${1:-ffg_pmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/bench_ff_geglu.py"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/p2 -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/p4 -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p3 -- $P > /dev/null 2>&1
{
  echo "K = 320 MFMA kernels (tools/bench_ff_geglu.py: [258048, 320] x [320, 2 x 1280] GEGLU, 320 -> 960, 320 -> 320; bf16 + f16), rocprofv3 --pmc, mean per dispatch"
  python3 $R/tools/pmc_summary.py $OUT/p1 ff_geglu
  python3 $R/tools/pmc_summary.py $OUT/p2 ff_geglu
  python3 $R/tools/pmc_summary.py $OUT/p4 ff_geglu
  f=$(find $OUT/p3 -name "*kernel_stats.csv" | head -1)
  grep ff_geglu $f | head -6
} > $OUT/pmc_ff_geglu.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
cat $OUT/pmc_ff_geglu.txt
