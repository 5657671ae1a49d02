#!/bin/bash
# Round profile: rocprofv3 kernel stats of the bench command per workload, the PMC traffic passes for the dominant
# kernels (separate --pmc runs, scripts/pmc_traffic_passes.sh), then the plain bench lines of every workload.
#   usage (on the GPU box): bash scripts/profile_round.sh <tag> <commit>
TAG=${1:-prof}
COMMIT=${2:-unknown}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 $R/bench.py --no-cpu --no-extras "$@" > $OUT/bench_${name}_under_rocprof.json 2> $OUT/stats_$name.log
  cp $(ls $OUT/stats_$name/*/*kernel_stats.csv | head -1) $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/stats_$name
}
stats c3 --steps 5 --warmup 2
stats c2 --workload c2 --steps 10 --warmup 3
stats c4 --workload c4 --steps 3 --warmup 1
stats c5 --workload c5 --steps 5 --warmup 2
stats c3_clustered --workload c3-clustered --steps 3 --warmup 1
stats c4_clustered --workload c4-clustered --steps 3 --warmup 1
cd $R
for W in c3 c2 c4 c5 c3-clustered c4-clustered; do
  # (the whole workload: the launch shape -- queries per device batch -- must be the bench's own)
  bash scripts/pmc_traffic_passes.sh $TAG/pmc_$W --workload $W --no-extras --steps 1 --warmup 1 > /dev/null 2>&1
  python scripts/pmc_to_traffic.py gpurun_out/$TAG/pmc_$W $W $OUT/pmc_traffic.json > /dev/null
  rm -rf gpurun_out/$TAG/pmc_$W
done
python - <<PY
import json
d = json.load(open('$OUT/pmc_traffic.json'))
for w in d:
    d[w]['measured_at_commit'] = '$COMMIT'
json.dump(d, open('$OUT/pmc_traffic.json', 'w'), indent=1, sort_keys=True)
PY
cp $OUT/pmc_traffic.json profiles/pmc_summary.json
python bench.py --steps 20 --warmup 5 2> $OUT/bench_c3.log | tail -1 > $OUT/bench_c3.json
python bench.py --workload c2 --steps 20 --warmup 5 2> $OUT/bench_c2.log | tail -1 > $OUT/bench_c2.json
python bench.py --workload c4 --steps 3 --warmup 1 2> $OUT/bench_c4.log | tail -1 > $OUT/bench_c4.json
python bench.py --workload c5 --steps 5 --warmup 2 2> $OUT/bench_c5.log | tail -1 > $OUT/bench_c5.json
python bench.py --workload c3-clustered --steps 3 --warmup 1 --no-cpu 2> $OUT/bench_c3cl.log | tail -1 > $OUT/bench_c3cl.json
python bench.py --workload c4-clustered --steps 3 --warmup 1 --no-cpu 2> $OUT/bench_c4cl.log | tail -1 > $OUT/bench_c4cl.json
python scripts/r05_hybrid_probe.py > $OUT/hybrid_c3.json 2> $OUT/hybrid_c3.log
# the check the bench line's per-step fields promise: the dominant kernel's total time in the trace / the passes traced == the line's
# dominant_kernel_ms_per_step (the traced command runs steps + warmup host passes and steps + 1 resident ones)
python3 - <<PY
import csv, json
line = json.loads(open('$OUT/bench_c3_under_rocprof.json').read().strip().split('\n')[-1])
rows = list(csv.DictReader(open('$OUT/c3_kernel_stats.csv')))
gemm = [r for r in rows if 'k_jc69_gemm' in r['Name']]
tot = sum(float(r['TotalDurationNs']) for r in gemm) / 1e6
calls = sum(int(r['Calls']) for r in gemm)
steps, warm = line['steps'], line['warmup']
passes = (steps + warm) + (steps + 1)
r = line['roofline']
print('CHECK c3: k_jc69_gemm %d calls, %.2f ms in all = %.2f ms per pass over %d passes (host %d x %d calls + resident %d x %d batches = %d calls expected); the line under rocprof says dominant_kernel_ms_per_step = %.2f'
      % (calls, tot, tot / passes, passes, steps + warm, int(r['kernel_calls_per_step']), steps + 1, r['device_batches'],
         (steps + warm) * int(r['kernel_calls_per_step']) + (steps + 1) * r['device_batches'], r['dominant_kernel_ms_per_step']))
PY
bash scripts/r06_cl_poly_trace.sh > $OUT/cl_poly_trace.txt 2>&1
python scripts/shape_legs.py > $OUT/shape_legs.json 2> $OUT/shape_legs.log
for f in c3 c2 c4 c5 c3cl c4cl; do python3 -c "
import json; d=json.load(open('$OUT/bench_$f.json')); print('$f', round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline'].get('traffic'), d['resident']['per_kernel_ms_per_step'], d.get('cpu_baseline') and round(d['cpu_baseline']['value'],1))"; done
