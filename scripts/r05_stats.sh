#!/bin/bash
# kernel stats of a bench command (rocprofv3 --kernel-trace --stats); usage on the GPU box: bash scripts/r05_stats.sh <workload> <tag> [steps]
WL=${1:-c3}
TAG=${2:-r05_$WL}
STEPS=${3:-3}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --workload $WL --no-cpu --no-extras --steps $STEPS --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${WL}_kernel_stats.csv
rm -rf $OUT/stats
head -24 $OUT/${WL}_kernel_stats.csv | cut -c1-220
