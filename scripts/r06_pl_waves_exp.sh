#!/bin/bash
# wavefronts per SIMD of the bottom-up lean kernel's instantiation for trees with polytomies (368 bytes of scratch per lane at three)
#   bash scripts/r06_pl_waves_exp.sh > gpurun_out/r06_pl_waves_exp.txt
cd $GRAFT_REPO_ROOT
for FL in "" "-DLEAN_UP_WAVES_PL=2"; do
  APPLES_EXTRA_HIPCC_FLAGS="$FL" python -c "
import os
os.utime('apples_amd/csrc/sweep_lean.hip')
from apples_amd import build
build.build(verbose=False)" > /dev/null 2>&1
  echo "flags [$FL]"
  python scripts/shape_legs.py c3 c3-unrooted c3-polytomies c3-unrooted-clustered c3-polytomies-clustered 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.items(): print('  ', k, round(v['ms_per_step'],2), {a:round(b,2) for a,b in v['per_kernel_ms_per_step'].items()})"
done
