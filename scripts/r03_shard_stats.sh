# kernel stats of one 12 500-query shard of C3 (what one of eight ranks runs per step)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03shard}
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu --no-extras --queries 12500 --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/stats.log
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
