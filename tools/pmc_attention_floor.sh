#!/bin/bash
# Round 4, VERDICT item 5: is the shipped 8-wave attention kernel at its VALU-issue ceiling? Counters of the SHIPPED build at the
# two large shapes of the step (B 28 H 5 S 9216 and B 28 H 10 S 2304, D 64, bf16), separate rocprofv3 --pmc passes.
#   VALU issue share = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x cycles), cycles = GRBM_GUI_ACTIVE / 8 (summed over the XCDs)
#   matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CYCLES)   (per-SIMD busy cycles over per-CU... see the printout)
# Usage (through gpurun): tools/pmc_attention_floor.sh <tag>  -> gpurun_out/<tag>/pmc_attention.txt
TAG=${1:-attn_floor}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
BIN=$R/tools/attn_dev/attn_check
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for shape in bench1 bench2; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${shape}_a -- $BIN $shape > /dev/null 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/${shape}_b -- $BIN $shape > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${shape}_t -- $BIN $shape > $OUT/${shape}_run.txt 2>&1
done
python3 - <<PY > $OUT/pmc_attention.txt
import csv, glob, os
from collections import defaultdict
out = "$OUT"
def counters(d):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_flash8" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}
def duration(d):
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_flash8" in r["Name"]:
                return float(r["AverageNs"]) / 1e3, int(r["Calls"])
    return None, 0
print("shipped attn_flash8_kernel<bf16, exact scale>, rocprofv3 --pmc (two passes) + --kernel-trace per shape; mean per dispatch")
for shape, (B, H, S) in (("bench1", (28, 5, 9216)), ("bench2", (28, 10, 2304))):
    c = counters(os.path.join(out, shape + "_a")); c.update({k: v for k, v in counters(os.path.join(out, shape + "_b")).items() if k not in c})
    us, calls = duration(os.path.join(out, shape + "_t"))
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    flops = 4.0 * B * H * S * S * 64
    print(f"\nB {B} H {H} S {S} D 64: {us:.1f} us per launch ({calls} launches) = {flops / us / 1e6:.1f} TFLOP/s = {flops / us / 1e6 / 2500:.3f} of 2.5 PF")
    for k in sorted(c): print(f"    {k:28s} {c[k]:16.1f}")
    mfma_cycles = c["SQ_INSTS_MFMA"] * 8 * 4 / 4      # v_mfma_f32_32x32x16: 8 passes x 4 cycles = 32 cycles of its SIMD's matrix pipe... printed both ways below
    print(f"    cycles per XCD (GRBM_GUI_ACTIVE / 8)        {cycles:14.0f}  -> clock {cycles / us / 1e3:.3f} GHz")
    print(f"    VALU issue share   SQ_INSTS_VALU x 4 / (1024 SIMDs x cycles)          = {c['SQ_INSTS_VALU'] * 4 / (1024 * cycles):.3f}   (SQ_INSTS_VALU counts MFMA too: {c['SQ_INSTS_MFMA']:.0f} of them)")
    nonm = c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']
    print(f"    non-matrix VALU    (SQ_INSTS_VALU - SQ_INSTS_MFMA) x 4 / (1024 x cycles) = {nonm * 4 / (1024 * cycles):.3f}; per MFMA: {nonm / c['SQ_INSTS_MFMA']:.2f} VALU, {c.get('SQ_INSTS_VALU_TRANS', 0) / c['SQ_INSTS_MFMA']:.2f} transcendental")
    print(f"    matrix pipe        SQ_INSTS_MFMA x 32 cycles (32x32x16, 16-bit) / (1024 x cycles) = {c['SQ_INSTS_MFMA'] * 32 / (1024 * cycles):.3f}")
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c: print(f"    SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES) = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f} (per-SE aggregates: see MI355X_MICROARCH.md for the normalisation)")
PY
rm -rf $OUT/bench1_a $OUT/bench1_b $OUT/bench1_t $OUT/bench2_a $OUT/bench2_b $OUT/bench2_t
cat $OUT/pmc_attention.txt
