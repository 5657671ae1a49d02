# Sweep kernels compiled for more resident wavefronts per SIMD (fewer registers each)
for w in 2 3 4; do
  rm -f apples_amd/csrc/sweep.o
  APPLES_EXTRA_HIPCC_FLAGS="-DAPPLES_SWEEP_WAVES=$w" python -m apples_amd.build > /dev/null 2>&1
  echo "== min waves/SIMD $w: $(python bench.py --steps 3 --warmup 1 --no-cpu --timed resident 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['per_kernel_ms_per_step'])") teams 4096: $(APPLES_SWEEP_TEAMS=4096 python bench.py --steps 3 --warmup 1 --no-cpu --timed resident 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['per_kernel_ms_per_step']['sweep_ms'])")"
done
rm -f apples_amd/csrc/sweep.o; python -m apples_amd.build > /dev/null 2>&1
