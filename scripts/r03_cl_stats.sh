# kernel stats of the clustered route (rocprofv3 --kernel-trace --stats), top kernels printed
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03cl3}
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu --no-extras --workload c3-clustered --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/stats.log
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
cut -d, -f1-4 $OUT/kernel_stats.csv | cut -c1-150 | head -16
