# kernel trace of one C3-size batch (8192 queries): where the selection time goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c3trace -- python3 $R/scripts/c3_probe.py 8192 > $R/gpurun_out/c3trace.log 2>&1 < /dev/null
f=$(ls $R/gpurun_out/c3trace/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cut -c1-140 "$f" | head -14; else tail -5 $R/gpurun_out/c3trace.log; fi
