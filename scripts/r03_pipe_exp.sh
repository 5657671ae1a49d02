# Round 3: does the distance pass of batch i+1 beside selection + sweep of batch i pay with the lean sweep?  (round 2: no)
one() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident "$@" 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['resident']['ms_per_step'],2), d['resident']['per_kernel_ms_per_step'])"; }
echo "== default: $(one)"
echo "== APPLES_PIPELINE=2: $(APPLES_PIPELINE=2 one)"
echo "== APPLES_PIPELINE=2 APPLES_GEMM_CUS=224: $(APPLES_PIPELINE=2 APPLES_GEMM_CUS=224 one)"
echo "== APPLES_PIPELINE=2 APPLES_GEMM_CUS=192: $(APPLES_PIPELINE=2 APPLES_GEMM_CUS=192 one)"
echo "== APPLES_PIPELINE=4: $(APPLES_PIPELINE=4 one)"
