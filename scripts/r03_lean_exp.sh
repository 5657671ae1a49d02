# Round 3: the lean sweep (sweep_lean.hip: bottom-up kernel + top-down kernel) against the level loop at C3
one() { timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident "$@" 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['resident']['per_kernel_ms_per_step'])"; }
echo "== level loop (APPLES_NO_SWEEP_LEAN=1): $(APPLES_NO_SWEEP_LEAN=1 one)"
echo "== lean: $(one)"
echo "== lean, bottom-up only: $(APPLES_SWEEP_DEBUG_PHASE=1 one)"
echo "== lean, big threshold 6144: $(APPLES_BIG_THRESHOLD=6144 one)"
echo "== lean, up 768 wgs (3/CU), down 768: $(APPLES_LEAN_UP_WGS=768 APPLES_LEAN_DOWN_WGS=768 one)"
echo "== lean, up 1536 wgs, down 1536: $(APPLES_LEAN_UP_WGS=1536 APPLES_LEAN_DOWN_WGS=1536 one)"
APPLES_LEAN_PROFILE=1 python bench.py --steps 2 --warmup 1 --no-cpu --no-extras --timed resident 2>&1 | grep "lean sweep"
