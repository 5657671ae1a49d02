#!/bin/bash
# where the lean sweep's wavefront-sized teams spend their cycles at config 3 (APPLES_LEAN_PROFILE: per-phase cycle shares on stderr)
cd $GRAFT_REPO_ROOT
APPLES_LEAN_PROFILE=1 python - <<'PY' 2>&1 | grep -i "lean sweep phases\|steps" | tail -6
import sys, numpy as np
sys.path.insert(0, '.')
from apples_amd import synth
from apples_amd.engine import Engine
d = synth.make_dataset(200000, 1000, 100000)
nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
e = Engine(d.tree, d.ref_seqs, nodes, method='OLS')
for _ in range(3): e.place_sequences(d.query_seqs)
print(e.timing())
e.close()
PY
