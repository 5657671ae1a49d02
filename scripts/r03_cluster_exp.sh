# Round 3: k_select_clusters at C3 size (4 972 clusters): member words in flight per lane
one() { timeout 600 python bench.py --workload c3-clustered --steps 3 --warmup 1 --no-cpu --no-extras --timed resident 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['resident']['per_kernel_ms_per_step'])"; }
echo "== built default (unroll 4): $(one)"
for u in 1 2 8; do
  rm -f apples_amd/csrc/select.o
  APPLES_EXTRA_HIPCC_FLAGS="-DCLUSTER_UNROLL=$u" python -m apples_amd.build > /dev/null 2>&1
  echo "== unroll $u: $(one)"
done
rm -f apples_amd/csrc/select.o; python -m apples_amd.build > /dev/null 2>&1
