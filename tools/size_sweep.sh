#!/bin/bash
# Rasterizer fwd+bwd throughput over workload sizes (run on the GPU box): one bench.py JSON line per size.
# Every run is bounded by its own timeout; a failing size stops the sweep (no further GPU work after a fault).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/size_sweep.jsonl
: > $OUT
run() { timeout -k 10 240 python $R/bench.py --path raster --no-cpu-baseline --steps 10 --warmup 2 --gaussians $1 --width $2 --height $3 >> $OUT 2>> $R/gpurun_out/size_sweep.err; }
run 100000 800 800 && run 500000 1920 1080 && run 1500000 1920 1080 && run 3000000 1920 1080 && run 3000000 3840 2160 && run 6000000 3840 2160 && run 12000000 3840 2160
rc=$?
python3 - <<PY
import json
for l in open("$OUT"):
    l = l.strip()
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    print(d["config"].get("workload", d["config"]), "->", d["value"], d["unit"], f'{d["ms_per_step"]:.3f} ms/step')
PY
exit $rc
