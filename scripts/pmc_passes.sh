#!/bin/bash
# rocprofv3 PMC passes over one bench run (separate passes: TCC has 4 slots, FETCH_SIZE takes 3).
# usage: bash scripts/pmc_passes.sh <outdir> [bench args...]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/$OUT/$name -- python3 $R/bench.py --no-cpu "${BENCH_ARGS[@]}" > $R/gpurun_out/$OUT/$name.log 2>&1; }
BENCH_ARGS=("$@")
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcc2 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
ls $R/gpurun_out/$OUT
