#!/bin/bash
# The STAMPED diagnostic build of csrc/linear_n320.hip, generated from the product source by tools/n320_dev/stamps.patch (never a second
# hand-kept copy: if the product file moved away from the patch, this fails and says so). Wave 0 of every block leaves s_memrealtime /
# s_memtime at the ends of its prologue, main loop and epilogue + its hardware id in a buffer of its own; the launcher prints the mean
# phase lengths and the per-CU gaps to stderr (tools/experiments/n320_stamps.py drives it; profiles/round5_n320_phased_ab.txt).
#   tools/n320_dev/build_stamped.sh [extra hipcc flags, e.g. -DLN3_LOADERS=8]   ->  ab/n320_stamped.so   (MVI_HIP_LIB=ab/n320_stamped.so python ...)
set -e
R=$(cd $(dirname $0)/../.. && pwd)
cd $R && python3 -m multiview_inpaint_amd.build > /dev/null
mkdir -p ab/n320_stamped
cp multiview_inpaint_amd/csrc/linear_n320.hip ab/n320_stamped/linear_n320.hip
patch -s ab/n320_stamped/linear_n320.hip tools/n320_dev/stamps.patch || { echo "tools/n320_dev/stamps.patch no longer applies to csrc/linear_n320.hip"; exit 1; }
sed -i 's|"../../include/|"|; s|"unet_io.h"|"'$R'/multiview_inpaint_amd/csrc/unet_io.h"|' ab/n320_stamped/linear_n320.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -fno-slp-vectorize -Wno-unused-result \
  -DLN3_STAMPS=1 "$@" -I$R/include -I$R/multiview_inpaint_amd/csrc -c ab/n320_stamped/linear_n320.hip -o ab/n320_stamped/linear_n320.o
OBJS=$(ls multiview_inpaint_amd/csrc/_obj/*.o | grep -v "/linear_n320.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS ab/n320_stamped/linear_n320.o -o ab/n320_stamped.so
echo built ab/n320_stamped.so
