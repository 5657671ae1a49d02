#!/bin/bash
# per-kernel time of the clustered route against the device batch's size (the launches' floors): 4 passes of config 3 through clusters
# at batches of 33 334 / 16 672 / 8 352 queries, rocprofv3 --kernel-trace --stats of each
#   bash scripts/r06_floor_exp.sh > gpurun_out/r06_floor_exp.txt
R=$GRAFT_REPO_ROOT
WL=${1:-c3-clustered}
cd /tmp && export TMPDIR=/tmp
for MB in 0 16672 8352; do
  rm -rf /tmp/fl_$MB
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fl_$MB -- python3 $R/scripts/r06_floor_exp.py $MB $WL > /dev/null 2> /tmp/fl_$MB.log
  grep max_batch /tmp/fl_$MB.log
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/fl_$MB/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:16]:
    print('  %-66s calls %5s per pass %7.3f ms avg %7.3f ms' % (r['Name'][:66], r['Calls'], float(r['TotalDurationNs'])/4e6, float(r['AverageNs'])/1e6))
PY
done
