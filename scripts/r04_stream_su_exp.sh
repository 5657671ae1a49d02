# k_select_stream: 16-byte loads in flight per lane (SU2_LOADS) and wavefronts per SIMD (STREAM_WAVES) against config 5's
# selection time (4 096-row block, 12 500-row shard).  Rebuilds select.o on the box per variant.
cd $GRAFT_REPO_ROOT
one() { for Q in 4096 12500; do python bench.py --workload c5 --queries $Q --no-cpu --no-extras --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   rows $Q: select %.3f ms, step %.3f ms' % (d['roofline']['per_kernel_ms_per_step']['select_ms'], d['ms_per_step']))"; done; }
for flags in "" "-DSU2_LOADS=6" "-DSU2_LOADS=8" "-DSTREAM_WAVES=3" "-DSTREAM_WAVES=3 -DSU2_LOADS=8" "-DSTREAM_WAVES=2 -DSU2_LOADS=8"; do
  rm -f apples_amd/csrc/select.o
  APPLES_EXTRA_HIPCC_FLAGS="$flags" python -m apples_amd.build > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== [$flags]"; one
done
rm -f apples_amd/csrc/select.o; python -m apples_amd.build > /dev/null 2>&1
