#!/usr/bin/env python3
"""Config 5's 4 096-row block: what k_select_stream's time is made of.  The same table with -b 25 (the bench's: some rows need the
top-up rule and are read twice, the second time by the slower refill loop) and with -b 1 (no row needs it: one streaming pass per
row), the selection phase's device time each, and how many rows the rule fired for."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from apples_amd import synth
from apples_amd.engine import Engine
n_leaves, Q = 200000, int(os.environ.get('NQ', 4096))
ds = synth.make_dataset(n_leaves, 4, Q)
nodes = np.array([ds.tree.name_to_node[n] for n in ds.ref_names], np.int32)
index = synth.TreeIndex(ds.tree)
rs = np.random.default_rng(3)
q_leaf = rs.integers(0, n_leaves, size=Q); q_pend = rs.exponential(0.01, size=Q)
D = np.empty((Q, n_leaves))
for lo in range(0, Q, 2048):
    hi = min(Q, lo + 2048)
    D[lo:hi] = synth.fast_distance_rows(ds.tree, index, q_leaf, q_pend, list(range(lo, hi)), seed_noise=7 + lo)
inside = ((D >= 0) & (D <= 0.2)).sum(axis=1)
res = {'rows': Q, 'rows_with_fewer_than_25_inside_the_threshold': int((inside < 25).sum())}
for b in (25, 1):
    eng = Engine(ds.tree, None, method='BME', criterion='MLSE', threshold=0.2, baseobs=b)
    h, _ = eng.upload_table(D, nodes)
    eng.place_resident(h)
    sel = sw = 0.0
    for _ in range(5):
        eng.place_resident(h)
        t = eng.timing(); sel += t['select_ms']; sw += t['sweep_ms']
    res['b_%d' % b] = {'select_ms': sel / 5, 'sweep_ms': sw / 5, 'TBps': Q * n_leaves * 8 / (sel / 5 * 1e-3) / 1e12}
    eng.close()
print(json.dumps(res))
