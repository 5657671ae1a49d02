#!/bin/bash
# kernel stats of the C4 bench command (rocprofv3 --kernel-trace --stats); usage on the GPU box: bash scripts/r04_c4_stats.sh <tag>
TAG=${1:-r04c4}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --workload c4 --no-cpu --no-extras --steps 3 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/c4_kernel_stats.csv
rm -rf $OUT/stats
head -20 $OUT/c4_kernel_stats.csv | cut -c1-200
