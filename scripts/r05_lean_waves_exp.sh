#!/bin/bash
# Wavefronts per SIMD of the wavefront-sized sweep teams' two kernels (registers against occupancy): usage on the GPU box:
#   bash scripts/r05_lean_waves_exp.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_lean_waves
mkdir -p $OUT
cd $R
for v in "" "-DLEAN_UP_WAVES=4" "-DLEAN_DOWN_WAVES=4" "-DLEAN_UP_WAVES=4 -DLEAN_DOWN_WAVES=4" "-DLEAN_UP_WAVES=2 -DLEAN_DOWN_WAVES=2"; do
  rm -f apples_amd/csrc/sweep_lean.o
  APPLES_EXTRA_HIPCC_FLAGS="$v" python -m apples_amd.build > /dev/null 2>&1
  for w in c3 c5 c2; do
    python bench.py --workload $w --no-cpu --no-extras --steps 6 --warmup 2 2> /dev/null | tail -1 > $OUT/line.json
    echo "[$v] $w: $(python -c "
import json; d=json.load(open('$OUT/line.json')); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['resident']['per_kernel_ms_per_step'].items()})")" | tee -a $OUT/summary.txt
  done
done
rm -f apples_amd/csrc/sweep_lean.o; python -m apples_amd.build > /dev/null 2>&1
