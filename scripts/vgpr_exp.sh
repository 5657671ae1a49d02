for w in 1 3 4; do
  rm -f apples_amd/csrc/sweep.o
  APPLES_EXTRA_HIPCC_FLAGS="-DAPPLES_SWEEP_WAVES=$w" python -m apples_amd.build > /dev/null 2>&1
  echo "== min waves/SIMD $w"
  python bench.py --steps 5 --warmup 2 --no-cpu 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['per_kernel_ms_per_step'])"
done
rm -f apples_amd/csrc/sweep.o; python -m apples_amd.build > /dev/null 2>&1
