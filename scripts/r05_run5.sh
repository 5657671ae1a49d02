mkdir -p gpurun_out/r05e
(time timeout 1200 python -m pytest tests/test_gpu_fuzz.py -q -s -x -k "clustered" 2>&1 | tail -12) > gpurun_out/r05e/fuzz.log 2>&1
bash scripts/r05_stats.sh c3-clustered r05e 2 > gpurun_out/r05e/stats.txt 2>&1
timeout 600 python bench.py --workload c3-clustered --no-cpu --no-extras --steps 3 --warmup 1 > gpurun_out/r05e/c3cl.json 2> gpurun_out/r05e/c3cl.err
cat gpurun_out/r05e/fuzz.log; cut -c1-150 gpurun_out/r05e/stats.txt | head -16; python - <<'PY'
import json
j=json.load(open('gpurun_out/r05e/c3cl.json'))
print(j['value'], j['ms_per_step'], j['roofline']['per_kernel_ms_per_step'])
PY
