# config 5, resident block: one pass as it is against APPLES_TABLE_PIPELINE=k sub-batches (selection of i + 1 beside the sweep of i)
cd $GRAFT_REPO_ROOT
for Q in 4096 12500; do
  for K in 0 2 4; do
    APPLES_TABLE_PIPELINE=$K python bench.py --workload c5 --no-cpu --queries $Q --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rows $Q sub-batches $K: %.0f q/s, %.2f ms per pass' % (d['value'], d['ms_per_step']), {k: round(v, 2) for k, v in d['roofline']['per_kernel_ms_per_step'].items()})"
  done
done
