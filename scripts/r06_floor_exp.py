#!/usr/bin/env python3
"""The clustered route's fixed cost per device batch, kernel by kernel: config 3 through clusters with the device batch capped at
`max_batch` queries (0: the workspace's own), 4 passes; run under rocprofv3 --kernel-trace --stats by scripts/r06_floor_exp.sh.

    python scripts/r06_floor_exp.py <max_batch>
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from apples_amd import synth  # noqa: E402
from apples_amd.engine import Engine  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 0
name = sys.argv[2] if len(sys.argv) > 2 else 'c3-clustered'
n_leaves, L, Q, protein, method, thr = bench.WORKLOADS[name]
ds = synth.make_dataset(n_leaves, L, Q, protein=protein)
nodes = np.array([ds.tree.name_to_node[n] for n in ds.ref_names], np.int32)
clusters = bench.make_clusters(ds, thr, protein) if name.endswith('-clustered') else None
eng = Engine(ds.tree, ds.ref_seqs, nodes, clusters=clusters, protein=protein, method=method, criterion='MLSE', threshold=thr, baseobs=25,
             overlap=0.001, device=0, max_batch=mb)
q = np.ascontiguousarray(ds.query_seqs[:Q])
eng.place_sequences(q)
t = []
for _ in range(3):
    t0 = time.perf_counter()
    eng.place_sequences(q)
    t.append((time.perf_counter() - t0) * 1e3)
print('max_batch %d: batch %s, %.2f ms per pass (min of 3), phases %s' % (mb, eng.describe().get('batch'), min(t),
      {k: round(v, 2) for k, v in eng.timing().items() if k.endswith('_ms')}), file=sys.stderr)
