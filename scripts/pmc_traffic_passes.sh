#!/bin/bash
# The three PMC passes profiles/pmc_summary.json needs (HBM bytes per launch), for any bench workload.
# usage: bash scripts/pmc_traffic_passes.sh <outdir under gpurun_out> [bench args...]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
BENCH_ARGS=("$@")
run() { name=$1; shift; timeout 280 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/$OUT/$name -- python3 $R/bench.py --no-cpu "${BENCH_ARGS[@]}" > $R/gpurun_out/$OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcc2 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum
ls $R/gpurun_out/$OUT
