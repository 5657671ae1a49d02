#!/bin/bash
# What the libm squares of the residual (sweep_math.h:pow2_libm, x ** 2 with libm's bits) cost the sweeps: the same builds with plain
# squares (results differ in the last place now and then: timing only).  usage on the GPU box: bash scripts/r05_plain_sq_exp.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_plain_sq.txt
: > $OUT
cd $R
for v in "" "-DSWEEP_EXP_PLAIN_SQ"; do
  rm -f apples_amd/csrc/sweep_lean.o apples_amd/csrc/sweep.o apples_amd/csrc/sweep_scan.o
  APPLES_EXTRA_HIPCC_FLAGS="$v" python -m apples_amd.build > /dev/null 2>&1
  for w in c3-clustered c4-clustered c3 c5; do
    echo "[$v] $w: $(python bench.py --workload $w --steps 4 --warmup 1 --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['resident']['per_kernel_ms_per_step'].items()})")" | tee -a $OUT
  done
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/psq
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psq -- python3 $R/bench.py --workload c3-clustered --no-cpu --no-extras --steps 3 --warmup 1 > /dev/null 2>&1
  echo "[$v] $(grep -h "k_blocks_down\|k_blocks_up\|k_lean_down\|k_lean_up" /tmp/psq/*/*kernel_stats.csv | cut -d, -f1,4 | sed 's/(anonymous namespace):://; s/void //' | tr '\n' ' ')" | tee -a $OUT
  cd $R
done
rm -f apples_amd/csrc/sweep_lean.o apples_amd/csrc/sweep.o apples_amd/csrc/sweep_scan.o; python -m apples_amd.build > /dev/null 2>&1
