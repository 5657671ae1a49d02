#!/bin/bash
# timing experiments on the scoredist candidate kernel (APPLES_SD_DBG: 1 = every piece re-reads piece 0, 2 = no lookups,
# 4 = the set-up alone (prefix of the segment counts), 8 = the rounds without the evaluation)
cd $GRAFT_REPO_ROOT
for D in 0 3 4 8; do
  echo "== APPLES_SD_DBG=$D"
  cd /tmp && export TMPDIR=/tmp
  APPLES_SD_DBG=$D timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sdexp_$D -- python3 $GRAFT_REPO_ROOT/bench.py --workload c4 --no-cpu --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob('/tmp/sdexp_$D/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k_sd_exact' in r['Name'] or 'k_sd_topup' in r['Name']:
        print('%-40s avg %8.3f ms' % (r['Name'][:40], float(r['AverageNs']) / 1e6))
PY
done
