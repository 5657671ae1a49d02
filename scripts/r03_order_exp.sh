# Round 3: what is locality between neighbouring queries worth?  (queries in the tree order of their sister leaves)
one() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident "$@" 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['resident']['ms_per_step'],2), d['resident']['per_kernel_ms_per_step'])"; }
echo "== c3, random order: $(one)"
echo "== c3, tree order: $(APPLES_BENCH_TREE_ORDER=1 one)"
echo "== clustered, random order: $(one --workload c3-clustered)"
echo "== clustered, tree order: $(APPLES_BENCH_TREE_ORDER=1 one --workload c3-clustered)"
echo "== c4, random order: $(one --workload c4)"
echo "== c4, tree order: $(APPLES_BENCH_TREE_ORDER=1 one --workload c4)"
