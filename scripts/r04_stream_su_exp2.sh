# k_select_stream, second round: fewer, deeper streams?  Loads in flight per lane (SU2_LOADS, compile time) x workgroups on the chip
# (APPLES_STREAM_GRID: 1 024 = four per CU) with the 640-entry queues (30 KB per workgroup whatever SU2_LOADS is).
cd $GRAFT_REPO_ROOT
one() { for Q in 4096 12500; do for G in 1024 768 512 256; do APPLES_STREAM_GRID=$G python bench.py --workload c5 --queries $Q --no-cpu --no-extras --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   rows $Q grid $G: select %.3f ms, step %.3f ms' % (d['roofline']['per_kernel_ms_per_step']['select_ms'], d['ms_per_step']))"; done; done; }
for flags in "-DSU2_LOADS=8" "-DSU2_LOADS=4" "-DSU2_LOADS=8 -DSTREAM_WAVES=2"; do
  rm -f apples_amd/csrc/select.o
  APPLES_EXTRA_HIPCC_FLAGS="$flags" python -m apples_amd.build > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== [$flags]"; one
done
rm -f apples_amd/csrc/select.o; python -m apples_amd.build > /dev/null 2>&1
