# merge layout: where to hand queries to workgroup-sized teams (which still use the node map)
one() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --workload c3 --timed resident 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['per_kernel_ms_per_step'])"; }
echo "== map: $(one)"
export APPLES_SWEEP_MERGE=1
echo "== merge: $(one)"
for t in 8192 16384 40000; do echo "== merge, big threshold $t: $(APPLES_BIG_THRESHOLD=$t one)"; done
