#!/bin/bash
# Experiment build of the K = 320 GEGLU / plain projection kernel (timing ablations MVI_FFG_EXPERIMENT=<n>: 1 no GELU arithmetic, 2 no
# stores, 3 no DMA, 4 no LDS fragment reads, 5 no per-step barrier, 6 one wave per SIMD; -DMVI_FFG_STAMPS: in-kernel cycle stamps):
# the production objects of libmvi_hip.so with tools/ff_dev/ff_geglu_x.hip (the kernel WITH its experiment modes, wrong results on
# purpose in modes 1 - 6) compiled under -DMVI_FFG_EXPERIMENTS in place of csrc/ff_geglu.hip, written to tools/ff_dev/x/libmvi_hip.so
# (never into the package). Use:  MVI_HIP_LIB=tools/ff_dev/x/libmvi_hip.so MVI_FFG_EXPERIMENT=2 python tools/bench_ff_geglu.py
set -e
R=$(cd $(dirname $0)/../.. && pwd)
cd $R && python3 -m multiview_inpaint_amd.build > /dev/null
mkdir -p tools/ff_dev/x
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -fno-slp-vectorize \
  -DMVI_FFG_EXPERIMENTS $FFG_DEFS -Wno-unused-variable -Wno-unused-but-set-variable -Iinclude -c tools/ff_dev/ff_geglu_x.hip -o tools/ff_dev/x/ff_geglu.o
OBJS=$(ls multiview_inpaint_amd/csrc/_obj/*.o | grep -v ff_geglu.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS tools/ff_dev/x/ff_geglu.o -o tools/ff_dev/x/libmvi_hip.so
echo built tools/ff_dev/x/libmvi_hip.so
