#!/bin/bash
# is k_blocks_up / k_blocks_down bound by its longest tile?  the same kernels at fewer queries per device batch
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_blk_tail
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for nq in 1000 4000 14000; do
  rm -rf $OUT/stats
  NQ=$nq timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/scripts/r05_c3cl_probe.py > $OUT/probe_$nq.json 2> $OUT/log.txt
  echo "== nq $nq: $(grep -h "k_blocks_up\|k_blocks_down\|k_lean_up\|k_lean_down\|k_select_clusters<3, 512" $OUT/stats/*/*kernel_stats.csv | cut -d, -f1,2,4 | sed 's/(anonymous namespace):://;s/void //' | tr '\n' ' ')" | tee -a $OUT/summary.txt
done
