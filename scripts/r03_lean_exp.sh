# Round 3: the lean sweep at C3 (singleton and clustered routes): routing threshold and workgroup-team size
one() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident "$@" 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['resident']['per_kernel_ms_per_step'])"; }
echo "== lean: $(one)"
echo "== lean, 512-thread big teams: $(APPLES_LEAN_BIG_TEAM=512 one)"
echo "== lean, big threshold 3072: $(APPLES_BIG_THRESHOLD=3072 one)"
echo "== lean, big threshold 6144: $(APPLES_BIG_THRESHOLD=6144 one)"
echo "== lean, big threshold 6144, 512-thread big teams: $(APPLES_BIG_THRESHOLD=6144 APPLES_LEAN_BIG_TEAM=512 one)"
echo "== clustered: $(one --workload c3-clustered)"
echo "== clustered, big threshold 8192: $(APPLES_BIG_THRESHOLD=8192 one --workload c3-clustered)"
echo "== clustered, big threshold 8192, 512-thread big teams: $(APPLES_BIG_THRESHOLD=8192 APPLES_LEAN_BIG_TEAM=512 one --workload c3-clustered)"
