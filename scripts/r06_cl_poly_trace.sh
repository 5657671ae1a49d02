#!/bin/bash
# kernel times of the clustered route on config 3's tree and on the same tree with 1 % polytomies (rocprofv3 --kernel-trace --stats)
#   bash scripts/r06_cl_poly_trace.sh [leg ...] > gpurun_out/r06_cl_poly_trace.txt     (legs of scripts/shape_legs.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for LEG in ${@:-c3-clustered c3-polytomies-clustered}; do
  rm -rf /tmp/tr_$LEG
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$LEG -- python3 $R/scripts/shape_legs.py $LEG > /dev/null 2> /tmp/tr_$LEG.log
  echo "== $LEG"
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/tr_$LEG/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:18]:
    print('%-70s calls %5s total %8.2f ms avg %8.3f ms' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6))
PY
done
