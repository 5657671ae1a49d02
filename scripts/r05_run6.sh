mkdir -p gpurun_out/r05f
python -m pytest tests/test_gpu_scale.py -q -x -k "clade_blocks" 2>&1 | tail -5 > gpurun_out/r05f/t.log
python -m pytest tests/test_gpu_shapes.py -q -x -k "scan_sweep or clustered_route or scoredist_fused" 2>&1 | tail -5 >> gpurun_out/r05f/t.log
python scripts/r05_c4_clustered_probe.py > gpurun_out/r05f/c4cl.log 2>&1
APPLES_PROBE_DEBUG=no_blocks python scripts/r05_c4_clustered_probe.py >> gpurun_out/r05f/c4cl.log 2>&1
cat gpurun_out/r05f/t.log gpurun_out/r05f/c4cl.log
