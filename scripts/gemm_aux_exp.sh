# Cache policy of the GEMM-form distance kernel's DMA loads
one() { timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --workload c3 --timed resident 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['per_kernel_ms_per_step']['dist_ms'])"; }
for a in 0 1 2 3; do
  rm -f apples_amd/csrc/dist_gemm.o
  APPLES_EXTRA_HIPCC_FLAGS="-DGM_AUX=$a" python -m apples_amd.build > /dev/null 2>&1 || { echo "build failed: $a"; continue; }
  echo "== aux $a: $(one)"
done
rm -f apples_amd/csrc/dist_gemm.o; python -m apples_amd.build > /dev/null 2>&1
