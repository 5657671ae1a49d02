# Per-kernel times of the C3 pass with the process restricted to m of every 4 CUs: which kernels are bound by
# per-CU resources (time ~ 4/m) and which by chip-wide ones (time flat)?
one() { timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --workload c3 --timed resident 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['per_kernel_ms_per_step'])"; }
for nib in f 7 3 1; do
  mask=0x$(python3 -c "print('$nib'*64)")
  echo "== ROC_GLOBAL_CU_MASK nibble $nib: $(ROC_GLOBAL_CU_MASK=$mask one)"
done
echo "== HSA_CU_MASK 0:0-127: $(HSA_CU_MASK=0:0-127 one)"
