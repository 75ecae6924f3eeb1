#!/bin/bash
# A/B of attention kernel build variants on the GPU box: tools/ab_attention.sh "<flags A>" "<flags B>" ...
# Each variant rebuilds attn_flash.hip with MVI_ATTN_FLAGS and runs the quick micro-benchmark.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for flags in "$@"; do
  echo "=== MVI_ATTN_FLAGS='$flags'" | tee -a gpurun_out/ab_attention.log
  rm -f multiview_inpaint_amd/csrc/_obj/attn_flash.o
  MVI_ATTN_FLAGS="$flags" python -m multiview_inpaint_amd.build >> gpurun_out/ab_attention.log 2>&1
  python tools/bench_attention.py --quick 2>&1 | tee -a gpurun_out/ab_attention.log
done
rm -f multiview_inpaint_amd/csrc/_obj/attn_flash.o
python -m multiview_inpaint_amd.build >> gpurun_out/ab_attention.log 2>&1
