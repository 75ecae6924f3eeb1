#!/bin/bash
# Offline selection of the library GEMM solutions for the SVD step (run on the MI355X box via gpurun):
#   tools/tune_svd_gemms.sh [ms per shape, default 100] -> gpurun_out/tunableop_gfx950.csv
# Copy the result to multiview_inpaint_amd/svd/tunableop_gfx950.csv; bench_svd.enable_gemm_tuning() looks it up.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/tunableop_tune*.csv
MVI_SVD_GEMM_TUNING_MS=${1:-100} MVI_SVD_GEMM_TUNING_ITERS=${2:-30} MVI_SVD_GEMM_TUNING_OUT=$PWD/gpurun_out/tunableop_tune.csv \
  python tools/experiments/svd_tunable.py 1 2>&1 | grep -v amdgpu.ids | tail -2
cp gpurun_out/tunableop_tune0.csv gpurun_out/tunableop_gfx950.csv 2>/dev/null || cp gpurun_out/tunableop_tune.csv gpurun_out/tunableop_gfx950.csv
wc -l gpurun_out/tunableop_gfx950.csv
# check: a second process that only looks the solutions up
cp gpurun_out/tunableop_gfx950.csv multiview_inpaint_amd/svd/tunableop_gfx950.csv
python tools/experiments/svd_tunable.py 1 2>&1 | grep -v amdgpu.ids | tail -1
MVI_SVD_TUNED_GEMMS=0 python tools/experiments/svd_tunable.py 1 2>&1 | grep -v amdgpu.ids | tail -1
