#!/bin/bash
# The STAMPED build of the two 8-wave attention kernels — generated from the product sources by the sed recipe below, never kept as a
# second copy: around the key loop of every block, wave 0 / lane 0 takes s_memtime and s_memrealtime and
# writes the two differences PAST THE END of the output tensor (16 bytes per block; the harness allocates them: attn_check clock).
# Output: ab/attn_stamped.so = the production objects with csrc/attn_flash8.hip and csrc/attn_flash8m16.hip replaced.
#   tools/attn_dev/build_stamped.sh && MVI_HIP_LIB=... (or LD_LIBRARY_PATH=ab/stamped) tools/attn_dev/attn_check clock
set -e
R=$(cd $(dirname $0)/../.. && pwd)
cd $R && python3 -m multiview_inpaint_amd.build > /dev/null
mkdir -p ab/stamped
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -fno-slp-vectorize -Wno-unused-result"
for f in attn_flash8 attn_flash8m16; do
  sed -e 's|^    run(std::false_type{});$|    const uint64_t st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();\n    run(std::false_type{});\n    if (tid == 0) { uint64_t* sp = reinterpret_cast<uint64_t*>(out + (int64_t)(total_blocks / (q_blocks * H)) * Sq * o_rs) + 2 * blockIdx.x; sp[0] = __builtin_amdgcn_s_memtime() - st_c0; sp[1] = __builtin_amdgcn_s_memrealtime() - st_r0; }|' \
      -e 's|"../../include/|"|' multiview_inpaint_amd/csrc/$f.hip > ab/stamped/$f.hip
  grep -q st_c0 ab/stamped/$f.hip || { echo "the recipe no longer matches $f.hip"; exit 1; }
  /opt/rocm/bin/hipcc $FLAGS -I$R/include -c ab/stamped/$f.hip -o ab/stamped/$f.o
done
OBJS=$(ls multiview_inpaint_amd/csrc/_obj/*.o | grep -v "/attn_flash8.o" | grep -v "/attn_flash8m16.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS ab/stamped/attn_flash8.o ab/stamped/attn_flash8m16.o -o ab/stamped/libmvi_hip.so
echo built ab/stamped/libmvi_hip.so
