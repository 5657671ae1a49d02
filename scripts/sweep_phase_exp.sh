# Bottom-up share of the C3 sweep (APPLES_SWEEP_DEBUG_PHASE=1 stops after the bottom-up pass; results are then garbage)
one() { timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --workload c3 --timed resident 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['per_kernel_ms_per_step'])"; }
echo "== full: $(one)"
echo "== bottom-up only: $(APPLES_SWEEP_DEBUG_PHASE=1 one)"
