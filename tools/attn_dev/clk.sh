#!/bin/bash
# cycles per dispatch (GRBM_GUI_ACTIVE / 8) and duration for attention experiment variants: tools/attn_dev/clk.sh <xp> ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for x in "$@"; do
  rm -rf /tmp/clk_$x
  MVI_ATTN_FOLD_SCALE=1 MVI_ATTN_EXPERIMENT=$x MVI_ATTN_VARIANT=8 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clk_$x -- $R/tools/attn_dev/attn_check bench1 > /dev/null 2>&1
  python3 - $x <<'PY'
import csv,glob,sys
x=sys.argv[1]
cyc=[];dur=[]
for f in glob.glob(f'/tmp/clk_{x}/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attn_flash8' in r['Kernel_Name'] and r['Counter_Name']=='GRBM_GUI_ACTIVE': cyc.append(float(r['Counter_Value'])/8)
for f in glob.glob(f'/tmp/clk_{x}/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attn_flash8' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
import statistics as st
if cyc and dur: print(f"xp {x}: cycles {st.mean(cyc)/1e6:.3f} M  duration {st.mean(dur):.3f} ms  clock {st.mean(cyc)/st.mean(dur)/1e6:.3f} GHz")
else: print('xp',x,'no data',len(cyc),len(dur))
PY
done
