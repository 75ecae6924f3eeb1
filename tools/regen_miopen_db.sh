#!/bin/bash
# Re-records MIOpen's user find-db for the convolution problems of the SVD step on the MI355X box (run through gpurun): one
# step with an EMPTY MIOPEN_USER_DB_PATH and cudnn.benchmark on, then the files MIOpen wrote are copied to gpurun_out/<tag>/
# (merge them into multiview_inpaint_amd/svd/miopen_userdb/ by hand: same file names).
TAG=${1:-miopen_db}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT /tmp/mvi_miopen_regen
rm -f /tmp/mvi_miopen_regen/*
export MIOPEN_USER_DB_PATH=/tmp/mvi_miopen_regen
cd $R; T0=$(date +%s)
MVI_SVD_MIOPEN_FIND=1 python bench.py --path svd --no-cpu-baseline --svd-steps 2 > $OUT/bench_first.json 2> $OUT/bench_first.err
echo "first run: $(( $(date +%s) - T0 )) s"; cp /tmp/mvi_miopen_regen/* $OUT/ 2>/dev/null
ls -la $OUT
tail -2 $OUT/bench_first.err
