#!/bin/bash
# Per-kernel durations of the GroupNorm forms at one shape at a time (rocprofv3 --kernel-trace --stats around tools/bench_groupnorm.py)
cd /tmp && export TMPDIR=/tmp
for shp in 28,320,72,128 28,640,72,128 28,640,36,64 28,1280,18,32; do
  export GN_SHAPE=$shp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gnp_$shp -- python3 /root/repo/tools/bench_groupnorm.py > /dev/null 2>&1
  echo "== $shp"
  f=$(find /tmp/gnp_$shp -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'mvi::' in r['Name']: print('%-58s calls %4s  avg %8.1f us' % (r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e3))
"
done
