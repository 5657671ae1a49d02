# Round 3: how fast is the distance GEMM with ONE four-wavefront workgroup per CU (half the registers left to other kernels)?
one() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident "$@" 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['resident']['ms_per_step'],2), d['resident']['per_kernel_ms_per_step'])"; }
echo "== default (QT 256, one eight-wavefront workgroup per CU): $(one)"
echo "== QT 128, two workgroups per CU: $(APPLES_GEMM_QT=128 one)"
echo "== QT 128, one workgroup per CU: $(APPLES_GEMM_QT=128 APPLES_GEMM_WGS_PER_CU=1 one)"
