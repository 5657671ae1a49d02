# Round 3: the lean sweep (sweep_lean.hip) against the level loop at C3, per wavefronts/SIMD and per pass
one() { timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident "$@" 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['resident']['per_kernel_ms_per_step'])"; }
echo "== level loop (APPLES_NO_SWEEP_LEAN=1): $(APPLES_NO_SWEEP_LEAN=1 one)"
for w in 2 3; do
  echo "== lean, $w waves/SIMD: $(APPLES_LEAN_WAVES=$w one)"
  echo "== lean, $w waves/SIMD, bottom-up only: $(APPLES_LEAN_WAVES=$w APPLES_SWEEP_DEBUG_PHASE=1 one)"
done
echo "== lean, big threshold 3072: $(APPLES_BIG_THRESHOLD=3072 one)"
echo "== lean, big threshold 6144: $(APPLES_BIG_THRESHOLD=6144 one)"
