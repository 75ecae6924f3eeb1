#!/bin/bash
# Runs on the MI355X box (through gpurun): two PMC passes over the rasterizer step for the binning kernels.
# Usage: tools/pmc_binning.sh <tag>
set -u
TAG=${1:-pmcbin}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --path raster --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -- $B > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $OUT/pmc_sq > $OUT/pmc_sq.txt
python3 $R/tools/pmc_summary.py $OUT/pmc_lds > $OUT/pmc_lds.txt
rm -rf $OUT/pmc_sq $OUT/pmc_lds
