#!/bin/bash
# One-rank RCCL step of the gradient-exchange forms on one box (the exchange's own kernels, launches and host work — everything but the
# wire): tools/dist_one_rank.sh <tag> -> gpurun_out/<tag>/dist_one_rank.txt. Round 5 adds the two forms of the support union (bit
# masks in one all-gather = the default; MVI_DIST_BYTE_MASK=1 = round 4's all-reduce(MAX) of a byte mask).
TAG=${1:-dist1}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
B="python bench.py --path raster --no-cpu-baseline --steps 20"
for rep in 1 2; do
  $B > $OUT/plain_$rep.json 2>/dev/null
  MVI_BENCH_FORCE_DIST=1 MVI_BENCH_EXCHANGE=compacted $B > $OUT/compacted_bits_$rep.json 2>/dev/null
  MVI_BENCH_FORCE_DIST=1 MVI_BENCH_EXCHANGE=compacted MVI_DIST_BYTE_MASK=1 $B > $OUT/compacted_bytes_$rep.json 2>/dev/null
  MVI_BENCH_FORCE_DIST=1 MVI_BENCH_EXCHANGE=factored $B > $OUT/factored_$rep.json 2>/dev/null
  MVI_BENCH_FORCE_DIST=1 MVI_BENCH_EXCHANGE=dense $B > $OUT/dense_$rep.json 2>/dev/null
done
python3 - <<PY > $OUT/dist_one_rank.txt
import json, glob, os
rows = {}
for f in sorted(glob.glob("$OUT/*_[12].json")):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    rows.setdefault(os.path.basename(f)[:-7], []).append(b)
print("Round 5: one-rank RCCL step of the exchange forms, same box, bench.py --path raster (20 steps, 1.5 M Gaussians, 1080p, sh 3), two runs each")
print("MVI_BENCH_FORCE_DIST=1 runs the exchange in an RCCL group of ONE rank: collectives move nothing, the difference to the plain step is the")
print("exchange's own kernels, launches and host work.\n")
base = min(b["ms_per_step"] for b in rows.get("plain", [{"ms_per_step": 0}]))
for k in ("plain", "compacted_bits", "compacted_bytes", "factored", "dense"):
    for b in rows.get(k, []):
        print(f"{k:18s} {b['value']:9.2f} Mpix/s  {b['ms_per_step']:.4f} ms per step  (+{b['ms_per_step'] - base:.4f} ms)  {b['config']['parallelism'][:150]}")
PY
cat $OUT/dist_one_rank.txt
