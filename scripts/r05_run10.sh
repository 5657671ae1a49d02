mkdir -p gpurun_out/r05j
(time python -m pytest tests/test_gpu_fuzz.py -q -x -k "clustered" 2>&1 | tail -3) > gpurun_out/r05j/t.log 2>&1
python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -q -x 2>&1 | tail -2 >> gpurun_out/r05j/t.log
for w in c3-clustered c4-clustered; do
  python bench.py --workload $w --no-cpu --no-extras --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['value']), round(d['ms_per_step'],2), d['roofline']['per_kernel_ms_per_step'], 'roofline', d['roofline']['kernel'], round(d['roofline']['frac'],3))" >> gpurun_out/r05j/t.log
done
cat gpurun_out/r05j/t.log
