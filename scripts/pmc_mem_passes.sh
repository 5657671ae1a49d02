#!/bin/bash
# Extra PMC passes for the memory pipeline (TA / TCP / TCC busy and stall counters); two counters
# per pass (more can exceed the hardware's slots for these blocks), every pass under its own timeout.
# usage: bash scripts/pmc_mem_passes.sh <outdir> [bench args...]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
run() { name=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/$OUT/$name -- python3 $R/bench.py --no-cpu "${BENCH_ARGS[@]}" > $R/gpurun_out/$OUT/$name.log 2>&1; echo "$name rc=$?"; }
BENCH_ARGS=("$@")
run ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum
run tcp1 TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
run tcc1 TCC_BUSY_sum TCC_TAG_STALL_sum
run tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run tcc3 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
ls $R/gpurun_out/$OUT
