# Round 3: where to cut between wavefront-sized (sweep_lean.hip) and workgroup-sized (sweep.hip) teams at C3
one() { timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident "$@" 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['resident']['per_kernel_ms_per_step'])"; }
for t in 2048 4096 8192 16384 40000; do
  echo "== big threshold $t: $(APPLES_BIG_THRESHOLD=$t one)"
done
echo "== big threshold 40000, 3 waves: $(APPLES_BIG_THRESHOLD=40000 APPLES_LEAN_WAVES=3 one)"
echo "== big threshold 8192, 3 waves: $(APPLES_BIG_THRESHOLD=8192 APPLES_LEAN_WAVES=3 one)"
