#!/bin/bash
# where the wavefront-sized sweep teams spend their cycles (APPLES_LEAN_PROFILE): config 3 singleton (NQ queries) and clustered
R=$GRAFT_REPO_ROOT
cd $R
echo "== c3 singleton, 25000 queries"; APPLES_LEAN_PROFILE=1 NQ=25000 python scripts/r05_hybrid_probe.py 2>&1 | grep -i "lean\|cycles\|steps" | head -8
echo "== c3 clustered, 20000 queries"; APPLES_LEAN_PROFILE=1 NQ=20000 python scripts/r05_c3cl_probe.py 2>&1 | grep -i "lean\|cycles\|steps" | head -4
