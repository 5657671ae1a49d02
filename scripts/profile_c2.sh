#!/bin/bash
# Round profile of the default bench (C2): rocprofv3 kernel stats of the bench command, the PMC
# passes, then the plain bench line (which picks the PMC traffic up from profiles/pmc_summary.json).
# usage (on the GPU box): bash scripts/profile_c2.sh <tag>      outputs under gpurun_out/<tag>/
TAG=${1:-prof}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu --steps 10 --warmup 3 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
cd $R
bash scripts/pmc_passes.sh $TAG/pmc --steps 3 --warmup 1 > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/$TAG/pmc > $OUT/pmc_summary.json
python scripts/pmc_to_traffic.py gpurun_out/$TAG/pmc c2 $OUT/pmc_traffic.json > /dev/null
cp $OUT/pmc_traffic.json profiles/pmc_summary.json.new 2>/dev/null
python - <<PY
import json
new = json.load(open('$OUT/pmc_traffic.json'))
p = 'profiles/pmc_summary.json'
allj = json.load(open(p))
allj.update(new)
json.dump(allj, open(p, 'w'), indent=1, sort_keys=True)
PY
python bench.py --steps 10 --warmup 3 2> $OUT/bench.log | tail -1 > $OUT/bench.json
tail -c 600 $OUT/bench.json
