# Round 3: the pipelined driver (sweep of batch i on a second stream while the front stream goes on with batch i + 1) with the
# slim workspace: do the sweep's tail and the distance pass's head fill one another's gaps?
one() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident "$@" 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['resident']['ms_per_step'],2), d['resident']['per_kernel_ms_per_step'], d['roofline']['launches_per_step'], d['config']['placed'])"; }
echo "== default: $(one)"
echo "== APPLES_PIPELINE=2: $(APPLES_PIPELINE=2 one)"
echo "== APPLES_PIPELINE=4: $(APPLES_PIPELINE=4 one)"
echo "== APPLES_PIPELINE=8: $(APPLES_PIPELINE=8 one)"
echo "== APPLES_PIPELINE=4 APPLES_GEMM_CUS=224: $(APPLES_PIPELINE=4 APPLES_GEMM_CUS=224 one)"
