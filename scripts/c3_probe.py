"""Exploratory C3-size run (200k leaves) with a reduced query count: kernel split + observed sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from apples_amd import synth
from apples_amd.engine import Engine
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
t = time.time(); ds = synth.make_dataset(200000, 1000, nq); print('synth s', time.time() - t, flush=True)
nodes = np.array([ds.tree.name_to_node[n] for n in ds.ref_names], np.int32)
eng = Engine(ds.tree, ds.ref_seqs, nodes, method='OLS')
h, n = eng.upload_queries(ds.query_seqs)
for i in range(3):
    t = time.time(); eng.place_resident(h); dt = time.time() - t
    print('pass', i, round(dt * 1e3, 2), {k: round(float(v), 2) for k, v in eng.timing().items()}, flush=True)
out = eng.fetch(h, n)
print(eng.describe())
print('n_obs mean/min/max', out['n_obs'].mean(), out['n_obs'].min(), out['n_obs'].max(), 'n_valid mean/max',
      out['n_valid'].mean(), out['n_valid'].max(), 'placed', (out['n_valid'] > 0).sum(), 'q/s', n / dt)
