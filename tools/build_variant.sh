#!/bin/bash
# An A/B build of libmvi_hip.so: the production objects with ONE source recompiled under extra flags, written to ab/<name>.so
# (never into the package):  tools/build_variant.sh <name> <source.hip> <extra hipcc flags ...>      then tools/ab_lib.sh <tag> ab/<name>.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
NAME=$1; SRC=$2; shift 2
cd $R && python3 -m multiview_inpaint_amd.build > /dev/null
mkdir -p ab/_obj_$NAME
BASE=$(basename $SRC .hip)
EXTRA=""
case $BASE in attn_flash|attn_flash8|attn_flash8m16|ff_geglu|linear_n320) EXTRA="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans";; esac
case $BASE in attn_flash8|attn_flash8m16|ff_geglu|linear_n320) EXTRA="$EXTRA -fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result $EXTRA "$@" -c multiview_inpaint_amd/csrc/$BASE.hip -o ab/_obj_$NAME/$BASE.o
OBJS=$(ls multiview_inpaint_amd/csrc/_obj/*.o | grep -v "/$BASE.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS ab/_obj_$NAME/$BASE.o -o ab/$NAME.so
echo built ab/$NAME.so
