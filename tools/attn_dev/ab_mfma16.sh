#!/bin/bash
# Same-box A/B of the 8-wave attention kernel on v_mfma_f32_32x32x16 (csrc/attn_flash8.hip, MVI_ATTN_MFMA16=0) against the same kernel
# on v_mfma_f32_16x16x32 (csrc/attn_flash8m16.hip, MVI_ATTN_MFMA16=1), both in the q-carries-the-scale form the SVD modules run:
#   1. correctness of the new kernel (attn_check's eight cases, exact and folded scale);
#   2. alternating timed runs at S = 9216 and 2304 on random data (~2 s of launches per shape and arm, three rounds);
#   3. in-kernel cycles and clock of both (stamped build, s_memtime / s_memrealtime), on random and on all-zero operands;
#   4. counters of both (separate --pmc passes).
# tools/attn_dev/ab_mfma16.sh <tag>   ->  gpurun_out/<tag>/ab_mfma16.txt        (needs tools/attn_dev/build_stamped.sh run beforehand)
TAG=${1:-ab_mfma16}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
BIN=$R/tools/attn_dev/attn_check
LOG=$OUT/ab_mfma16.txt
cd $R
{
echo "== 1. correctness, MVI_ATTN_MFMA16=1 (exact scale, then MVI_ATTN_FOLD_SCALE=1)"
for mm in 1 2; do MVI_ATTN_MFMA16=$mm MVI_ATTN_VARIANT=8 timeout -k 10 300 $BIN check 2>&1 | grep -v "^bench"; done
for mm in 1 2; do MVI_ATTN_MFMA16=$mm MVI_ATTN_VARIANT=8 MVI_ATTN_FOLD_SCALE=1 timeout -k 10 300 $BIN check 2>&1 | grep -v "^bench"; done
} > $LOG 2>&1
grep -q "CHECKS FAILED\|HIP error\|FAIL" $LOG && { echo "correctness failed: stopping"; cat $LOG; exit 1; }
{
echo "== 2. alternating timed runs (q carries the scale), random data"
for rep in 1 2 3; do for m in 0 1 2; do MVI_ATTN_MFMA16=$m MVI_ATTN_CHECK_QLOG2=1 timeout -k 10 120 $BIN benchlong | grep "^bench"; done; done
echo "== 3. stamped build: key-loop cycles and in-kernel clock, random then ZERO operands"
for mode in clock clock0; do for m in 0 1 2; do LD_LIBRARY_PATH=$R/ab/stamped MVI_ATTN_MFMA16=$m MVI_ATTN_CHECK_QLOG2=1 timeout -k 10 200 $BIN $mode | grep "^clock"; done; done
} >> $LOG 2>&1
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2; do
  export MVI_ATTN_MFMA16=$m MVI_ATTN_CHECK_QLOG2=1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1_$m -- $BIN bench1 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/p2_$m -- $BIN bench1 > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/p4_$m -- $BIN bench1 > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p3_$m -- $BIN bench1 > /dev/null 2>&1
  {
    echo "== 4. counters, MVI_ATTN_MFMA16=$m (attn_check bench1: B 28 H 5 S 9216, mean per dispatch)"
    python3 $R/tools/pmc_summary.py $OUT/p1_$m attn_flash
    python3 $R/tools/pmc_summary.py $OUT/p2_$m attn_flash
    python3 $R/tools/pmc_summary.py $OUT/p4_$m attn_flash
    f=$(find $OUT/p3_$m -name "*kernel_stats.csv" | head -1)
    grep attn_flash $f | head -2
  } >> $LOG 2>&1
  rm -rf $OUT/p1_$m $OUT/p2_$m $OUT/p3_$m $OUT/p4_$m
done
cat $LOG
