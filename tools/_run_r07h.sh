mkdir -p gpurun_out/r07h && timeout -k 10 600 python -m pytest tests/test_raster_gpu.py -m gpu -x -q -k "deferred or headline or forward_parity or raw or sparse" > gpurun_out/r07h/test.log 2>&1; echo "rc $?" >> gpurun_out/r07h/test.log; grep -v "^Extension modules" gpurun_out/r07h/test.log | tail -3
tools/ab_lib.sh r07h ab/lib_prev.so
for n in 0 512; do MVI_RASTER_FRONT_ENTRIES=$n python bench.py --path raster --no-cpu-baseline > gpurun_out/r07h/front$n.json 2>/dev/null; done
python3 - <<PY
import json
for n in ("front0","front512"):
    b=json.loads(open(f"gpurun_out/r07h/{n}.json").read().strip().splitlines()[-1])
    print(n,b["value"],b["ms_per_step"],{k:v["ms"] for k,v in b["stages"].items()})
PY
