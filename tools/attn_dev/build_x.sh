#!/bin/bash
# Experiment build of the attention kernels (timing ablations, in-kernel stamps, block timeline: MVI_ATTN_EXPERIMENT=<n>):
# the production objects of libmvi_hip.so with tools/attn_dev/attn_flash8_x.hip (the kernel WITH its experiment modes) compiled under -DMVI_ATTN_EXPERIMENTS in place of csrc/attn_flash8.hip, written to
# tools/attn_dev/x/libmvi_hip.so (never into the package). Use:  LD_LIBRARY_PATH=tools/attn_dev/x tools/attn_dev/attn_check ...
set -e
R=$(cd $(dirname $0)/../.. && pwd)
cd $R && python3 -m multiview_inpaint_amd.build > /dev/null
mkdir -p tools/attn_dev/x
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -fno-slp-vectorize \
  -DMVI_ATTN_EXPERIMENTS -Wno-unused-variable -Wno-unused-but-set-variable -Iinclude -c tools/attn_dev/attn_flash8_x.hip -o tools/attn_dev/x/attn_flash8.o
OBJS=$(ls multiview_inpaint_amd/csrc/_obj/*.o | grep -v attn_flash8.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS tools/attn_dev/x/attn_flash8.o -o tools/attn_dev/x/libmvi_hip.so
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/attn_dev/attn_check.cpp -Iinclude -Lmultiview_inpaint_amd/csrc -lmvi_hip -ldl \
  -Wl,-rpath,'$ORIGIN/../../multiview_inpaint_amd/csrc' -o tools/attn_dev/attn_check
echo built tools/attn_dev/x/libmvi_hip.so and tools/attn_dev/attn_check
