#!/bin/bash
# PMC traffic passes alone (see profile_round.sh).  usage: bash scripts/pmc_only.sh <tag> <commit>
TAG=$1; COMMIT=$2; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for W in c3 c2; do
  bash scripts/pmc_traffic_passes.sh $TAG/pmc_$W --workload $W --steps 1 --warmup 1 > /dev/null 2>&1
  python scripts/pmc_to_traffic.py gpurun_out/$TAG/pmc_$W $W $OUT/pmc_traffic.json > /dev/null
done
python - <<PY
import json
d = json.load(open('$OUT/pmc_traffic.json'))
for w in d:
    d[w]['measured_at_commit'] = '$COMMIT'
json.dump(d, open('$OUT/pmc_traffic.json', 'w'), indent=1, sort_keys=True)
for w in d:
    for k, v in d[w].items():
        if isinstance(v, dict): print(w, k, round(v['hbm_bytes_per_launch'] / 1e9, 2), 'GB', round(v['mean_ns_under_pmc'] / 1e6, 3), 'ms')
PY
