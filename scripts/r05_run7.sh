mkdir -p gpurun_out/r05g
(time python -m pytest tests/test_gpu_fuzz.py -q -s -x -k "clustered_scoredist" 2>&1 | tail -6) > gpurun_out/r05g/fuzz.log 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -x -k "protein or prot_" 2>&1 | tail -3 >> gpurun_out/r05g/fuzz.log
python scripts/r05_c4_clustered_probe.py > gpurun_out/r05g/c4cl.log 2>&1
cat gpurun_out/r05g/fuzz.log gpurun_out/r05g/c4cl.log
