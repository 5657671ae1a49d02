#!/bin/bash
# the short-form launch of the clustered selection's last phase (k_select_clusters SHORT_ONLY): entries its list holds against
# the workgroups a CU holds (1 024: five; 512: eight; 256: eight, more queries left to the general form)
#   bash scripts/r06_short_cap_exp.sh > gpurun_out/r06_short_cap_exp.txt
cd $GRAFT_REPO_ROOT
for CAP in ${CAPS:-1024 512 256}; do
  APPLES_EXTRA_HIPCC_FLAGS="-DSELECT_SHORT_ONLY_CAP=$CAP" python -c "
import os
os.utime('apples_amd/csrc/select.hip')
from apples_amd import build
build.build(verbose=False)" > /dev/null 2>&1
  echo "SELECT_SHORT_ONLY_CAP=$CAP"
  for R in 1 2; do timeout 300 python scripts/r06_floor_exp.py 0 2>&1 | tail -1; done
done
