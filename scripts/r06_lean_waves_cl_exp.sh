#!/bin/bash
# the lean sweep's wavefronts per SIMD on the CLUSTERED route (short lists above the clade blocks; round 5 measured the singleton route)
#   bash scripts/r06_lean_waves_cl_exp.sh > gpurun_out/r06_lean_waves_cl_exp.txt
cd $GRAFT_REPO_ROOT
for FL in "" "-DLEAN_UP_WAVES=4" "-DLEAN_DOWN_WAVES=4" "-DLEAN_UP_WAVES=4 -DLEAN_DOWN_WAVES=4" "-DLEAN_UP_WAVES=2 -DLEAN_DOWN_WAVES=2"; do
  APPLES_EXTRA_HIPCC_FLAGS="$FL" python -c "
import os
os.utime('apples_amd/csrc/sweep_lean.hip')
from apples_amd import build
build.build(verbose=False)" > /dev/null 2>&1
  echo "flags [$FL]"
  for R in 1 2; do timeout 300 python scripts/r06_floor_exp.py 0 2>&1 | tail -1; done
done
