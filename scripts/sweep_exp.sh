for t in 8192 5120 4096 3072 2048 1536 1024; do
  export APPLES_SWEEP_TEAMS=$t
  echo "== teams=$t"; python bench.py --steps 5 --warmup 2 --no-cpu 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['per_kernel_ms_per_step'])"
done
