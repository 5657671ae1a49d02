# Device-batch budget (GiB of batch buffers) against the C3 pass: fewer, larger launches
for g in 96 140 200; do
  echo "== APPLES_BATCH_GIB=$g: $(APPLES_BATCH_GIB=$g timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2), d['roofline']['launches_per_step'], d['resident']['per_kernel_ms_per_step'])")"
done
