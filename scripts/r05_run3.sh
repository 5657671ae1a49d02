mkdir -p gpurun_out/r05c
(time python -m pytest tests/test_gpu_shapes.py -q -x -k "c4_shape_clustered or other_criteria" 2>&1 | tail -30) > gpurun_out/r05c/shapes.log 2>&1
(time python -m pytest tests/test_gpu_cli.py tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_backbone.py -q -x 2>&1 | tail -15) > gpurun_out/r05c/rest.log 2>&1
cat gpurun_out/r05c/shapes.log gpurun_out/r05c/rest.log
