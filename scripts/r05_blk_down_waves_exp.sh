#!/bin/bash
# k_blocks_down compiled for 2 (default) / 3 wavefronts per SIMD, grid = that many workgroups per CU.  usage on the GPU box: bash scripts/r05_blk_down_waves_exp.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_blk_down_waves.txt
: > $OUT
cd $R
for v in "" "-DBLK_DOWN_WAVES=3"; do
  rm -f apples_amd/csrc/sweep_lean.o
  APPLES_EXTRA_HIPCC_FLAGS="$v" python -m apples_amd.build > /dev/null 2>&1
  for w in c3-clustered c4-clustered; do
    echo "[$v] $w: $(python bench.py --workload $w --steps 4 --warmup 1 --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['resident']['per_kernel_ms_per_step'].items()})")" | tee -a $OUT
  done
done
rm -f apples_amd/csrc/sweep_lean.o; python -m apples_amd.build > /dev/null 2>&1
