mkdir -p gpurun_out/r08c
timeout -k 10 300 python -m pytest tests/test_raster_gpu.py -m gpu -x -q -k "compacted or ranged or split or factored" > gpurun_out/r08c/test.log 2>&1; echo "rc $?" >> gpurun_out/r08c/test.log; grep -v "^Extension modules" gpurun_out/r08c/test.log | tail -15
python bench.py --path raster --no-cpu-baseline > gpurun_out/r08c/plain.json 2>/dev/null
MVI_BENCH_FORCE_DIST=1 MVI_BENCH_EXCHANGE=compacted python bench.py --path raster --no-cpu-baseline > gpurun_out/r08c/dist1_compacted.json 2>gpurun_out/r08c/dist1.err
MVI_BENCH_FORCE_DIST=1 MVI_BENCH_EXCHANGE=factored python bench.py --path raster --no-cpu-baseline > gpurun_out/r08c/dist1_factored.json 2>/dev/null
MVI_BENCH_FORCE_DIST=1 MVI_BENCH_EXCHANGE=dense python bench.py --path raster --no-cpu-baseline > gpurun_out/r08c/dist1_dense.json 2>/dev/null
python3 - <<PY
import json
for n in ("plain","dist1_compacted","dist1_factored","dist1_dense"):
    try:
        b=json.loads(open(f"gpurun_out/r08c/{n}.json").read().strip().splitlines()[-1])
        print(n,b["value"],b["ms_per_step"],b["config"]["parallelism"][:160])
    except Exception as e: print(n,"failed",e)
PY
tail -5 gpurun_out/r08c/dist1.err
