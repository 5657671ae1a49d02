# Round 3: where to cut between wavefront-sized and workgroup-sized lean teams at C3, and the workgroup team's size
one() { timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident "$@" 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['resident']['per_kernel_ms_per_step'])"; }
for t in 1024 2048 4096 8192; do
  echo "== big threshold $t: $(APPLES_BIG_THRESHOLD=$t one)"
  echo "== big threshold $t, 512-thread teams: $(APPLES_BIG_THRESHOLD=$t APPLES_LEAN_BIG_TEAM=512 one)"
done
echo "== big threshold 2048, 3 waves: $(APPLES_BIG_THRESHOLD=2048 APPLES_LEAN_WAVES=3 one)"
