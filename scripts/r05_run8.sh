mkdir -p gpurun_out/r05h
(time python -m pytest tests/test_gpu_fuzz.py -q -x -k "singleton or distance_table" 2>&1 | tail -4) > gpurun_out/r05h/fuzz.log 2>&1
for v in "" "APPLES_LEAN_TWO_KERNELS=1"; do
  for w in c3 c2 c5 c4 c3-clustered; do
    echo "== [$v] $w: $(env $v python bench.py --workload $w --no-cpu --no-extras --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2), d['roofline']['per_kernel_ms_per_step'], 'resident', round(d['resident']['ms_per_step'],2))")" >> gpurun_out/r05h/bench.log
  done
done
cat gpurun_out/r05h/fuzz.log gpurun_out/r05h/bench.log
