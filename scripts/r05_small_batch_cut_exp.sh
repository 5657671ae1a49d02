#!/bin/bash
# The routing cut (observed leaves above which a query goes to the workgroup-sized sweep teams) in small device batches: config 5's 4 096-row
# block and one 12 500-query shard of config 3.  APPLES_BIG_THRESHOLD fixes the cut for every batch size.  usage: bash scripts/r05_small_batch_cut_exp.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_small_batch_cut.txt
: > $OUT
cd $R
for cut in "" 4096 3072 2048 1536 1024; do
  e=""; [ -n "$cut" ] && e="APPLES_BIG_THRESHOLD=$cut"
  for args in "--workload c5 --steps 6 --warmup 2" "--queries 12500 --scaling weak --steps 8 --warmup 3" "--workload c2 --steps 8 --warmup 3"; do
    echo "[cut ${cut:-default}] [$args]: $(env $e python bench.py $args --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['resident']['per_kernel_ms_per_step'].items()})")" | tee -a $OUT
  done
done
