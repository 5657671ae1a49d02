for p in 1 2 4 8; do
  export APPLES_PIPELINE=$p
  echo "== pipeline=$p"; python bench.py --steps 5 --warmup 2 --no-cpu 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['per_kernel_ms_per_step'])"
done
unset APPLES_PIPELINE
python scripts/c3_probe.py 8192 2>&1 | tail -3 | head -1
