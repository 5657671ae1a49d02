"""Distance kernel at query tile T (roofline point T=1): packed reference streamed once per T queries."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from apples_amd import synth
from apples_amd.engine import Engine
n_leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ds = synth.make_dataset(n_leaves, 1000, nq)
nodes = np.array([ds.tree.name_to_node[n] for n in ds.ref_names], np.int32)
eng = Engine(ds.tree, ds.ref_seqs, nodes, method='OLS')
h, n = eng.upload_queries(ds.query_seqs)
info = None
for tile in (1, 4, 8, 16):
    for rep in range(3):
        eng.distances_resident(h, tile)
    t = eng.timing()
    info = eng.describe()
    packed = info['packed_bytes']
    ms = t['dist_ms']
    real = (packed * (n / tile) + n * info['n_rows'] * 8 + n * 48 * 8) / (ms * 1e-3) / 1e9   # reference re-read per tile + fp64 rows out
    algo = n * (info['n_rows'] * (1000 + 8) + 1000) / (ms * 1e-3) / 1e9
    print('tile', tile, 'ms', round(ms, 3), 'per query us', round(ms * 1e3 / n, 2), 'expected HBM GB/s', round(real), 'algorithmic GB/s', round(algo), flush=True)
print(info)
