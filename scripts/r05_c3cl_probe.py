#!/usr/bin/env python3
"""config 3's inputs through the default clustered route: one pass, then the block counters of the last device batch"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from apples_amd import synth
from apples_amd.engine import Engine
nq = int(os.environ.get('NQ', 100000))
ds = synth.make_dataset(200000, 1000, nq)
nodes = np.array([ds.tree.name_to_node[n] for n in ds.ref_names], np.int32)
eng = Engine(ds.tree, ds.ref_seqs, nodes, clusters=bench.make_clusters(ds, 0.2), method='OLS', threshold=0.2)
out = eng.place_sequences(ds.query_seqs)
t0 = time.perf_counter(); out = eng.place_sequences(ds.query_seqs); dt = time.perf_counter() - t0
d = eng.describe()
print(json.dumps({'ms': dt * 1e3, 'timing': eng.timing(), 'batch': d['batch'], 'blocks': d['cluster_blocks'], 'items_last': d['block_items_last_batch'],
                  'tiles_last': d['block_tiles_last_batch'], 'mean_obs': float(out['n_obs'].mean())}))
