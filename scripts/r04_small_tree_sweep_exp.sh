# bits-in-LDS level loop (default on small trees) against the merged lists + lean sweep (APPLES_SWEEP_MERGE=1) by backbone size
cd $GRAFT_REPO_ROOT
for N in 500 1000 2000 5000 10000 20000 24000; do
  for E in "" "APPLES_SWEEP_MERGE=1"; do
    env $E APPLES_BENCH_LEAVES=$N python bench.py --workload c2 --no-cpu --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('leaves $N [$E]: %.0f q/s, %.3f ms' % (d['value'], d['ms_per_step']), {k: round(v,3) for k,v in d['roofline']['per_kernel_ms_per_step'].items()})"
  done
done
