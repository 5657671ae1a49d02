#!/bin/bash
# k_blocks_up beside the last phase of the clustered selection: which of the two is launched first, on one stream or two, and how
# many persistent workgroups per CU k_blocks_up gets.  (APPLES_BLK_FIRST / APPLES_BLK_SERIAL: knobs of the library up to commit 27f4222.)
# usage on the GPU box: bash scripts/r05_blk_order_exp.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_blk_order.txt
: > $OUT
cd $R
for order in "" "APPLES_BLK_FIRST=1" "APPLES_BLK_SERIAL=1" "APPLES_BLK_SERIAL=1 APPLES_BLK_FIRST=1"; do
  for wgs in 6 4 3 2; do
    for w in c3-clustered c4-clustered; do
      line=$(env $order APPLES_BLK_UP_WGS=$wgs python bench.py --workload $w --steps 4 --warmup 1 --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['resident']['per_kernel_ms_per_step'].items()})")
      echo "[$order] up_wgs=$wgs $w: $line" | tee -a $OUT
    done
  done
done
