mkdir -p gpurun_out/r05b
python -m pytest tests/test_gpu_parity.py -q -s -k "protein_clustered" 2>&1 | tail -30 > gpurun_out/r05b/parity.log
(time python -m pytest tests/test_gpu_fuzz.py -q -s -k "scoredist" 2>&1 | tail -60) > gpurun_out/r05b/fuzz.log 2>&1
python scripts/r05_c4_clustered_probe.py > gpurun_out/r05b/c4cl.log 2>&1
APPLES_PROBE_DEBUG=no_fuse APPLES_PROBE_STEPS=1 python scripts/r05_c4_clustered_probe.py >> gpurun_out/r05b/c4cl.log 2>&1
cat gpurun_out/r05b/parity.log gpurun_out/r05b/fuzz.log gpurun_out/r05b/c4cl.log
