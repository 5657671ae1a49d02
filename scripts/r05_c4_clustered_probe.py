#!/usr/bin/env python3
"""Config 4's shape (50 000 leaves x 500 aa, 50 000 queries, FM) through the command line's default PROTEIN route: max-diameter
clusters at 1.2 x -f with consensus representatives of the 21-symbol alphabet.  Prints per-phase times of `steps` passes;
APPLES_PROBE_DEBUG = comma-separated debug switches (e.g. no_fuse), APPLES_PROBE_LEAVES / _QUERIES = another size."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from apples_amd import synth, treecluster  # noqa: E402
from apples_amd.engine import Engine  # noqa: E402
from apples_amd.fasta import Alignment  # noqa: E402
from apples_amd.reference import ReducedReference  # noqa: E402

n = int(os.environ.get('APPLES_PROBE_LEAVES', 50000))
Q = int(os.environ.get('APPLES_PROBE_QUERIES', 50000))
L = int(os.environ.get('APPLES_PROBE_L', 500))
thr = float(os.environ.get('APPLES_PROBE_F', 0.2))
dbg = tuple(x for x in os.environ.get('APPLES_PROBE_DEBUG', '').split(',') if x)
steps = int(os.environ.get('APPLES_PROBE_STEPS', 3))
ds = synth.make_dataset(n, L, Q, protein=True)
nodes = np.array([ds.tree.name_to_node[x] for x in ds.ref_names], np.int32)
t0 = time.perf_counter()
ca = ReducedReference(Alignment(ds.ref_names, ds.ref_seqs), True, treecluster.grouped(ds.tree, thr * 1.2)).cluster_arrays()
t_cl = time.perf_counter() - t0
eng = Engine(ds.tree, ds.ref_seqs, nodes, clusters=ca, protein=True, method='FM', threshold=thr, baseobs=25, debug=dbg)
out = eng.place_sequences(ds.query_seqs)
ph = {'dist_ms': 0.0, 'select_ms': 0.0, 'sweep_ms': 0.0, 'filter_ms': 0.0}
t0 = time.perf_counter()
for _ in range(steps):
    out = eng.place_sequences(ds.query_seqs)
    t = eng.timing()
    for k in ph:
        ph[k] += t[k]
dt = (time.perf_counter() - t0) / steps
info = eng.describe()
print(json.dumps({'debug': dbg, 'leaves': n, 'queries': Q, 'L': L, 'n_reps': info['n_reps'], 'cluster_fused': info.get('cluster_fused'),
                  'queries_per_s': Q / dt, 'ms_per_step': dt * 1e3, 'per_kernel_ms': {k: v / steps for k, v in ph.items()},
                  'mean_observed': float(np.mean(out['n_obs'])), 'placed': int((out['n_valid'] > 0).sum()),
                  'clustering_s': t_cl, 'crc': int(np.frombuffer(out.tobytes(), np.uint8).astype(np.uint64).sum())}))
