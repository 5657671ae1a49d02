#!/bin/bash
# rocprofv3 kernel stats of the C4 and C5 bench commands (profiles/r02_c4_kernel_stats.csv, r02_c5_kernel_stats.csv)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/more; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in c4 c5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- python3 $R/bench.py --workload $W --no-cpu --steps 3 --warmup 1 > $OUT/bench_${W}_under_rocprof.json 2> $OUT/stats_$W.log
  cp $(ls $OUT/stats_$W/*/*kernel_stats.csv | head -1) $OUT/${W}_kernel_stats.csv
  head -4 $OUT/${W}_kernel_stats.csv | cut -c1-150
done
