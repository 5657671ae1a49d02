one() { python bench.py --steps 5 --warmup 2 --no-cpu 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['per_kernel_ms_per_step'])"; }
rm -f apples_amd/csrc/dist.o
APPLES_EXTRA_HIPCC_FLAGS="-DMF_SKIP_EPILOGUE" python -m apples_amd.build > /dev/null 2>&1
echo "== main loop only"; one
rm -f apples_amd/csrc/dist.o; python -m apples_amd.build > /dev/null 2>&1
echo "== full"; one
