mkdir -p gpurun_out/r05a
python -m pytest tests/test_gpu_parity.py -q -k "protein_clustered" 2>&1 | tail -30 > gpurun_out/r05a/parity.log
python -m pytest tests/test_gpu_cli.py -q -k "prot_" 2>&1 | tail -30 > gpurun_out/r05a/cli.log
(time python -m pytest tests/test_gpu_fuzz.py -q -s 2>&1 | tail -80) > gpurun_out/r05a/fuzz.log 2>&1
python scripts/r05_c4_clustered_probe.py > gpurun_out/r05a/c4cl.log 2>&1
cat gpurun_out/r05a/parity.log gpurun_out/r05a/cli.log gpurun_out/r05a/fuzz.log gpurun_out/r05a/c4cl.log
