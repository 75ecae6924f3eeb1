#!/bin/bash
# Runs on the MI355X box (through gpurun): raster GPU tests, then rocprofv3 kernel stats + an un-profiled bench line.
# Usage: tools/prof_raster_quick.sh <tag> [notest]
set -u
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
if [ "${2:-}" != "notest" ]; then
  timeout -k 10 700 python -m pytest $R/tests/test_raster_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
  tail -2 $OUT/tests.log
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 10 --warmup 2 --path raster --no-cpu-baseline > $OUT/bench_under_trace.json 2>/dev/null
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/trace
cd $R && python bench.py --path raster --no-cpu-baseline > $OUT/bench.json 2>/dev/null
python3 - <<PY
import csv, json
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
tot=0
for r in rows:
    n=r['Name']
    if 'mvi::' in n:
        short=n.split('(')[0].replace('void ','')
        print(f"{short:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1000:8.1f} us")
j=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("Mpix/s", j["value"], "ms", j["ms_per_step"])
PY
