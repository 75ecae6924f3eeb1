#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate PMC passes as MI355X_MICROARCH.md prescribes) of path B's two hand-written
# contractions at their largest shapes of the 14 x 576x1024 step — the attention kernel (B 28, H 5, S 9216, D 64, bf16:
# tools/attn_dev/attn_check bench1) and the implicit-GEMM convolution (28 x 72x128, 640 -> 320: tools/experiments/conv3x3_n320_one.py)
# — so that roofline_svd_attention.traffic / roofline_svd_conv.traffic of the bench line stop being null.
# Run on the MI355X box via gpurun: tools/pmc_svd_traffic.sh <tag>  -> gpurun_out/<tag>/pmc_svd_traffic.txt, then
# python tools/make_svd_traffic_json.py gpurun_out/<tag> <tag>  -> profiles/svd_traffic.json
TAG=${1:-svd_traffic}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
BIN=$R/tools/attn_dev/attn_check
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
P="python3 $R/tools/experiments/conv3x3_n320_one.py"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/af -- $BIN bench1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/aw -- $BIN bench1 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/ah -- $BIN bench1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/at -- $BIN bench1 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cf -- $P > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cw -- $P > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/ch -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ct -- $P > /dev/null 2>&1
{
  echo "path B traffic passes (rocprofv3 --pmc, one counter family per pass, mean per dispatch; KiB for FETCH_SIZE / WRITE_SIZE)"
  echo "== attention: B 28 H 5 S 9216 D 64 bf16 (tools/attn_dev/attn_check bench1)"
  python3 $R/tools/pmc_summary.py $OUT/af attn_flash
  python3 $R/tools/pmc_summary.py $OUT/aw attn_flash
  python3 $R/tools/pmc_summary.py $OUT/ah attn_flash
  grep attn_flash $(find $OUT/at -name "*kernel_stats.csv" | head -1) | head -2
  echo "== implicit-GEMM convolution: 28 x 72x128, 640 -> 320, bf16 (tools/experiments/conv3x3_n320_one.py)"
  python3 $R/tools/pmc_summary.py $OUT/cf linear_n320
  python3 $R/tools/pmc_summary.py $OUT/cw linear_n320
  python3 $R/tools/pmc_summary.py $OUT/ch linear_n320
  grep linear_n320 $(find $OUT/ct -name "*kernel_stats.csv" | head -1) | head -2
} > $OUT/pmc_svd_traffic.txt
rm -rf $OUT/af $OUT/aw $OUT/ah $OUT/at $OUT/cf $OUT/cw $OUT/ch $OUT/ct
cat $OUT/pmc_svd_traffic.txt
