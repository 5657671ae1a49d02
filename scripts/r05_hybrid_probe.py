#!/usr/bin/env python3
"""Config 3's inputs with -c HYBRID (apples/Algorithm.py:83-91: the log2(num_nodes) smallest residuals, then the smallest pendant
among them): on the lean sweep (every edge kept in the entries, ranked after the top-down pass) and, with the `hybrid_records` switch,
through the level loop of sweep.hip with per-edge records (rounds 1 - 4).  One warm pass of 100 000
queries, host buffers -> placements in host memory, with the per-phase device times; ME and MLSE beside it."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from apples_amd import synth
from apples_amd.engine import Engine
nq = int(os.environ.get('NQ', 100000))
ds = synth.make_dataset(200000, 1000, nq)
nodes = np.array([ds.tree.name_to_node[n] for n in ds.ref_names], np.int32)
res = {}
for crit, dbg in (('MLSE', ()), ('ME', ()), ('HYBRID', ()), ('HYBRID', ('hybrid_records',))):
    eng = Engine(ds.tree, ds.ref_seqs, nodes, method='OLS', criterion=crit, threshold=0.2, debug=dbg)
    eng.place_sequences(ds.query_seqs)
    t0 = time.perf_counter(); out = eng.place_sequences(ds.query_seqs); dt = time.perf_counter() - t0
    t = eng.timing()
    res[crit + ('_records' if dbg else '')] = {'queries_per_s': nq / dt, 'ms_per_pass': dt * 1e3, 'dist_ms': t['dist_ms'], 'select_ms': t['select_ms'], 'sweep_ms': t['sweep_ms'],
                 'sweep_layout': eng.describe()['sweep_layout'], 'batch': eng.describe()['batch']}
    eng.close()
print(json.dumps(res))
