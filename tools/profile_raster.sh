#!/bin/bash
# Runs on the MI355X box (through gpurun): kernel-trace stats + separate PMC passes for the rasterizer
# step, written under gpurun_out/$1. Usage: tools/profile_raster.sh <tag>
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 1 --path raster --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/bench_under_trace.json 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/pmc_write -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_clk -- $B > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $OUT/pmc_clk > $OUT/pmc_clk.txt
python3 $R/tools/pmc_summary.py $OUT/pmc_fetch > $OUT/pmc_fetch.txt
python3 $R/tools/pmc_summary.py $OUT/pmc_write > $OUT/pmc_write.txt
python3 $R/tools/pmc_summary.py $OUT/pmc_sq > $OUT/pmc_sq.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_clk
# counters of THIS build first (the file carries the digest of the rasterizer sources), then the un-profiled bench line that quotes them
python3 $R/tools/make_traffic_json.py $OUT $TAG > $OUT/traffic_by_stage.json && cp $R/profiles/raster_traffic.json $OUT/raster_traffic.json
python3 $R/bench.py --steps 20 --warmup 3 --path raster > $OUT/bench.json 2> /dev/null
ls -la $OUT
