#!/bin/bash
# Records the MIOpen find-db entries and the TunableOp GEMM solutions of the SVD step in f16 (the reference's precision) on the
# MI355X box (run through gpurun): tools/tune_svd_f16.sh <tag> -> gpurun_out/<tag>/{*.udb.txt, *.ufdb.txt, tunableop_f16.csv}.
# Append the db lines to multiview_inpaint_amd/svd/miopen_userdb/* and the csv's GEMM lines to svd/tunableop_gfx950.csv.
TAG=${1:-f16_tune}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT /tmp/mvi_miopen_f16
rm -f /tmp/mvi_miopen_f16/*
cd $R
export MIOPEN_USER_DB_PATH=/tmp/mvi_miopen_f16
T0=$(date +%s)
MVI_SVD_MIOPEN_FIND=1 MVI_SVD_TUNED_GEMMS=0 python tools/experiments/svd_tunable.py 1 f16 2>&1 | grep -v amdgpu.ids | tail -1
echo "MIOpen search: $(( $(date +%s) - T0 )) s"; cp /tmp/mvi_miopen_f16/* $OUT/
T0=$(date +%s)
MVI_SVD_GEMM_TUNING_MS=100 MVI_SVD_GEMM_TUNING_ITERS=30 MVI_SVD_GEMM_TUNING_OUT=$OUT/tunableop_f16.csv python tools/experiments/svd_tunable.py 1 f16 2>&1 | grep -v amdgpu.ids | tail -1
echo "GEMM tuning: $(( $(date +%s) - T0 )) s"
ls -la $OUT
