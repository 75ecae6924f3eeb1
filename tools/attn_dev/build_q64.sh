#!/bin/bash
# Experiment build: libmvi_hip.so with tools/attn_dev/attn_q64.hip linked in place of csrc/attn_flash8.hip (written to tools/attn_dev/q64/,
# never into the package; Q64_X=<n> builds ablation n into tools/attn_dev/q64_x<n>/). Use:  LD_LIBRARY_PATH=tools/attn_dev/q64 tools/attn_dev/attn_check ...
set -e
R=$(cd $(dirname $0)/../.. && pwd)
cd $R && python3 -m multiview_inpaint_amd.build > /dev/null
OUT=tools/attn_dev/q64${Q64_X:+_x$Q64_X}${Q64_TAG:+_$Q64_TAG}
mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 ${Q64_FLAGS--mllvm -amdgpu-mfma-vgpr-form} -fno-honor-nans -fno-slp-vectorize \
  -Wno-unused-variable -Wno-unused-but-set-variable -Iinclude ${Q64_X:+-DQ64_X=$Q64_X} $Q64_DEFS -c tools/attn_dev/attn_q64.hip -o $OUT/attn_flash8.o
OBJS=$(ls multiview_inpaint_amd/csrc/_obj/*.o | grep -v attn_flash8.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $OUT/attn_flash8.o -o $OUT/libmvi_hip.so
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/attn_dev/attn_check.cpp -Iinclude -Lmultiview_inpaint_amd/csrc -lmvi_hip -ldl \
  -Wl,-rpath,'$ORIGIN/../../multiview_inpaint_amd/csrc' -o tools/attn_dev/attn_check
echo built $OUT/libmvi_hip.so
