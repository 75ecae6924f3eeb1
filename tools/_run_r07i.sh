mkdir -p gpurun_out/r07i
for ev in 1 2 3; do for n in 256; do MVI_RASTER_FRONT_EVERY=$ev MVI_RASTER_FRONT_ENTRIES=$n python bench.py --path raster --no-cpu-baseline > gpurun_out/r07i/every${ev}_$n.json 2>/dev/null; done; done
MVI_RASTER_FRONT_EVERY=2 MVI_RASTER_FRONT_ENTRIES=512 python bench.py --path raster --no-cpu-baseline > gpurun_out/r07i/every2_512.json 2>/dev/null
MVI_RASTER_FRONT_ENTRIES=0 python bench.py --path raster --no-cpu-baseline > gpurun_out/r07i/every0_0.json 2>/dev/null
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r07i/*.json")):
    b=json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f),b["value"],b["ms_per_step"],{k:v["ms"] for k,v in b["stages"].items()})
PY
tools/prof_raster_quick.sh r07i_prof notest
