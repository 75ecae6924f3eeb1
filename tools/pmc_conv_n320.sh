#!/bin/bash
# Matrix-pipe / VALU / LDS counters of the implicit-GEMM convolution (csrc/linear_n320.hip, kConv) at a level-0 shape, separate PMC
# passes (run on the MI355X box via gpurun): tools/pmc_conv_n320.sh <tag> -> gpurun_out/<tag>/pmc_conv_n320.txt
TAG=${1:-conv_pmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
P="python3 $R/tools/experiments/conv3x3_n320_one.py"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/p2 -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/p4 -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p3 -- $P > /dev/null 2>&1
{
  echo "implicit-GEMM convolution (tools/experiments/conv3x3_n320_one.py: 28 x 72x128, 640 -> 320, bf16: 1008 blocks, 90 chunks of K), rocprofv3 --pmc, mean per dispatch"
  python3 $R/tools/pmc_summary.py $OUT/p1 linear_n320
  python3 $R/tools/pmc_summary.py $OUT/p2 linear_n320
  python3 $R/tools/pmc_summary.py $OUT/p4 linear_n320
  f=$(find $OUT/p3 -name "*kernel_stats.csv" | head -1)
  grep linear_n320 $f | head -3
} > $OUT/pmc_conv_n320.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
cat $OUT/pmc_conv_n320.txt
