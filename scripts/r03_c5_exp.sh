# Round 3: k_select_stream at C5 (4 096 x 200 000 fp64 table rows resident): ring depth, occupancy, grid
one() { timeout 120 python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(r['per_kernel_ms_per_step']['select_ms'],3), 'ms, select GB/s', round(r['all_kernels_GBps'].get('table_select',0)), 'placed', d['config']['placed'])"; }
echo "== built default (ring 4, 4 waves): $(one)"
echo "== default build, grid 1024: $(APPLES_STREAM_GRID=1024 one)"
for cfg in "8 3" "6 4" "12 2"; do
  set -- $cfg
  rm -f apples_amd/csrc/select.o
  APPLES_EXTRA_HIPCC_FLAGS="-DSU2_LOADS=$1 -DSTREAM_WAVES=$2" python -m apples_amd.build > /dev/null 2>&1
  echo "== ring $1, $2 waves/SIMD: $(one)"
  echo "== ring $1, $2 waves/SIMD, grid 1024: $(APPLES_STREAM_GRID=1024 one)"
done
rm -f apples_amd/csrc/select.o; python -m apples_amd.build > /dev/null 2>&1
