# Round 3: the lean sweep (sweep_lean.hip) against the level loop at C3, per wavefronts/SIMD and per pass
one() { timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['resident']['per_kernel_ms_per_step'])"; }
echo "== level loop (APPLES_NO_SWEEP_LEAN=1): $(APPLES_NO_SWEEP_LEAN=1 one)"
for w in 2 3 4; do
  echo "== lean, $w waves/SIMD: $(APPLES_LEAN_WAVES=$w one)"
done
echo "== lean 2 waves, pass A only: $(APPLES_SWEEP_DEBUG_PHASE=1 one)"
echo "== lean 2 waves, passes A+B: $(APPLES_SWEEP_DEBUG_PHASE=2 one)"
echo "== lean 3 waves, pass A only: $(APPLES_LEAN_WAVES=3 APPLES_SWEEP_DEBUG_PHASE=1 one)"
echo "== lean 3 waves, passes A+B: $(APPLES_LEAN_WAVES=3 APPLES_SWEEP_DEBUG_PHASE=2 one)"
echo "== lean 3 waves, teams 3072: $(APPLES_LEAN_WAVES=3 APPLES_SWEEP_TEAMS=3072 one)"
echo "== lean 3 waves, teams 6144: $(APPLES_LEAN_WAVES=3 APPLES_SWEEP_TEAMS=6144 one)"
echo "== lean 4 waves, teams 4096: $(APPLES_LEAN_WAVES=4 APPLES_SWEEP_TEAMS=4096 one)"
