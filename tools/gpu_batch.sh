#!/bin/bash
# Runs a list of GPU steps one after the other on the MI355X box (through gpurun), each under its own time limit, logging to
# gpurun_out/<tag>/<name>.log. A step that is killed at its limit (exit 124 / 137) ends the batch: nothing else is started on a GPU
# that may be wedged. A step that merely fails (a red test) does not.
#   tools/gpu_batch.sh <tag> "<name>|<seconds>|<command>" ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (limit ${secs}s): $cmd" | tee -a $OUT/batch.log
  t0=$(date +%s)
  timeout -k 10 $secs bash -c "$cmd" > $OUT/$name.log 2>&1
  rc=$?
  echo "== $name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a $OUT/batch.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name was killed at its limit: batch stopped" | tee -a $OUT/batch.log; exit 1; fi
done
exit 0
