# Round 3: the lean sweep at C3 (singleton and clustered routes): routing threshold with the finely graded queue
one() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident "$@" 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['resident']['per_kernel_ms_per_step'])"; }
for t in 4096 8192 12288 16384 24576 40000; do
  echo "== big threshold $t: $(APPLES_BIG_THRESHOLD=$t one)"
done
for t in 8192 12288 16384 24576; do
  echo "== clustered, big threshold $t: $(APPLES_BIG_THRESHOLD=$t one --workload c3-clustered)"
done
