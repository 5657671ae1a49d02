mkdir -p gpurun_out/r05i
(time python -m pytest tests/test_gpu_shapes.py -q -x -k "c3_shape_200k or c5_shape" 2>&1 | tail -4) > gpurun_out/r05i/t.log 2>&1
python -m pytest tests/test_gpu_scale.py -q -x 2>&1 | tail -3 >> gpurun_out/r05i/t.log
python bench.py --no-cpu 2>gpurun_out/r05i/bench.err | tail -1 > gpurun_out/r05i/bench_halves.json
APPLES_NO_LEAN_HALVES=1 python bench.py --no-cpu 2>>gpurun_out/r05i/bench.err | tail -1 > gpurun_out/r05i/bench_nohalves.json
cat gpurun_out/r05i/t.log
python - <<'PY'
import json
for f in ('halves','nohalves'):
    d=json.load(open('gpurun_out/r05i/bench_%s.json'%f))
    p=d['strong_scaling_proxy']; o=d['other_workloads']
    print(f, round(d['ms_per_step'],2), 'shards', [round(x,2) for x in p['ms_shard']], 'pred', round(p['predicted_speedup_at_8'],2), 'c2', round(o['c2']['ms_per_step'],3), 'c5', round(o['c5']['ms_per_step'],3), o['c5']['per_kernel_ms_per_step'], 'c5shard', round(o['c5_shard_12500_rows']['ms_per_step'],3), 'c4', round(o['c4']['ms_per_step'],2), 'c4cl', round(o['c4_clustered']['ms_per_step'],2), 'clustered', round(d['clustered']['ms_per_step'],2))
PY
