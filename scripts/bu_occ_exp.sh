# bottom-up-only build of the sweep at several occupancies (timing experiment)
one() { python bench.py --steps 5 --warmup 2 --no-cpu 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['per_kernel_ms_per_step'])"; }
rm -f apples_amd/csrc/sweep.o
APPLES_EXTRA_HIPCC_FLAGS="-DAPPLES_BU_ONLY" python -m apples_amd.build > /dev/null 2>&1
for t in 1024 2048 3072 4096; do echo "== BU only, teams=$t"; APPLES_SWEEP_TEAMS=$t one; done
rm -f apples_amd/csrc/sweep.o
APPLES_EXTRA_HIPCC_FLAGS="-DAPPLES_BU_ONLY -DAPPLES_SWEEP_WAVES=5" python -m apples_amd.build > /dev/null 2>&1
echo "== BU only, 5 waves/SIMD, teams=5120"; APPLES_SWEEP_TEAMS=5120 one
rm -f apples_amd/csrc/sweep.o; python -m apples_amd.build > /dev/null 2>&1
