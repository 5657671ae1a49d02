# Round 3: what bounds the C3 sweep -- teams in flight, CUs, or the memory system?  ("What's weak" 4)
# (a) resident workgroups per CU limited by LDS padding at a constant grid of 768 workgroups (3 072 teams)
# (b) the same team counts through APPLES_SWEEP_TEAMS (smaller grid)
# (c) CU masks with the default grid
one() { timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['resident']['per_kernel_ms_per_step'])"; }
echo "== default: $(one)"
echo "== LDS pad 70000 (1 workgroup = 4 teams per CU resident, 1 024 in all; grid 768): $(APPLES_SWEEP_LDS_PAD=70000 one)"
echo "== LDS pad 40000 (2 per CU by LDS as by registers): $(APPLES_SWEEP_LDS_PAD=40000 one)"
echo "== APPLES_SWEEP_TEAMS=1024 (grid 256): $(APPLES_SWEEP_TEAMS=1024 one)"
echo "== APPLES_SWEEP_TEAMS=2048 (grid 512): $(APPLES_SWEEP_TEAMS=2048 one)"
echo "== APPLES_SWEEP_TEAMS=512 (grid 128): $(APPLES_SWEEP_TEAMS=512 one)"
echo "== APPLES_SWEEP_TEAMS=512 APPLES_SWEEP_BIG_WGS=128: $(APPLES_SWEEP_TEAMS=512 APPLES_SWEEP_BIG_WGS=128 one)"
for nib in 7 3 1; do
  mask=0x$(python3 -c "print('$nib'*64)")
  echo "== ROC_GLOBAL_CU_MASK nibble $nib: $(ROC_GLOBAL_CU_MASK=$mask one)"
  echo "== ROC_GLOBAL_CU_MASK nibble $nib + LDS pad 70000: $(ROC_GLOBAL_CU_MASK=$mask APPLES_SWEEP_LDS_PAD=70000 one)"
done
