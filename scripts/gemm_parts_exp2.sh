# Is the DMA cost of the GEMM-form distance kernel latency (the wait before the barrier) or issue/bandwidth?
one() { timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu --workload c3 --timed resident 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['per_kernel_ms_per_step']['dist_ms'])"; }
for flags in "-DGM_SKIP_EPILOGUE" "-DGM_SKIP_EPILOGUE -DGM_NO_VMWAIT"; do
  rm -f apples_amd/csrc/dist_gemm.o
  APPLES_EXTRA_HIPCC_FLAGS="$flags" python -m apples_amd.build > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== [$flags] $(one)"
done
rm -f apples_amd/csrc/dist_gemm.o; python -m apples_amd.build > /dev/null 2>&1
