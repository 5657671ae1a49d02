# Round 3: clustered route, the workgroup-sized teams' size and the routing threshold (its largest queries observe most of the tree)
one() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident "$@" 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['resident']['ms_per_step'],2), d['resident']['per_kernel_ms_per_step'])"; }
echo "== clustered, teams of 256: $(one --workload c3-clustered)"
echo "== clustered, teams of 512: $(APPLES_LEAN_BIG_TEAM=512 one --workload c3-clustered)"
echo "== clustered, teams of 512, threshold 16384: $(APPLES_LEAN_BIG_TEAM=512 APPLES_BIG_THRESHOLD=16384 one --workload c3-clustered)"
echo "== clustered, teams of 256, threshold 4096: $(APPLES_BIG_THRESHOLD=4096 one --workload c3-clustered)"
echo "== c3, teams of 512: $(APPLES_LEAN_BIG_TEAM=512 one)"
