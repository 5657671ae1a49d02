mkdir -p gpurun_out/r05d
(time timeout 1200 python -m pytest tests/test_gpu_fuzz.py -q -s -x -k "clustered" 2>&1 | tail -40) > gpurun_out/r05d/fuzz.log 2>&1
timeout 600 python bench.py --workload c3-clustered --no-cpu --no-extras --steps 3 --warmup 1 > gpurun_out/r05d/c3cl.json 2> gpurun_out/r05d/c3cl.err
APPLES_NO_BLOCKS=1 timeout 600 python bench.py --workload c3-clustered --no-cpu --no-extras --steps 3 --warmup 1 > gpurun_out/r05d/c3cl_noblk.json 2>> gpurun_out/r05d/c3cl.err
cat gpurun_out/r05d/fuzz.log; cut -c1-1800 gpurun_out/r05d/c3cl.json; echo; cut -c1-1800 gpurun_out/r05d/c3cl_noblk.json; tail -5 gpurun_out/r05d/c3cl.err
