#!/bin/bash
# Same-box A/B of two builds of libmvi_hip.so on the rasterizer bench: tools/ab_lib.sh <tag> <A.so> [<B.so> ...]
# (the shipped library is always measured too, as "shipped"). Libraries must live inside the repo snapshot (e.g. ab/).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for rep in 1 2; do
  for lib in shipped "$@"; do
    name=$(basename $lib .so)
    if [ "$lib" = shipped ]; then env -u MVI_HIP_LIB python bench.py --path raster --no-cpu-baseline > $OUT/${name}_$rep.json 2>/dev/null
    else MVI_HIP_LIB=$R/$lib python bench.py --path raster --no-cpu-baseline > $OUT/${name}_$rep.json 2>/dev/null; fi
  done
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/*.json")):
    b=json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), b["value"], b["ms_per_step"], {k:v["ms"] for k,v in b["stages"].items()})
PY
