#!/bin/bash
# k_blocks_up beside the selection's last phase: workgroups per CU of its persistent grid
R=$GRAFT_REPO_ROOT
for w in 2 3 4 6; do
  echo "== APPLES_BLK_UP_WGS=$w: $(APPLES_BLK_UP_WGS=$w python $R/scripts/r05_c3cl_probe.py | cut -c1-200)"
done
