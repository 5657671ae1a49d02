"""Is the C3 sweep's time set by a few queries with huge observed sets (a tail), or by the bulk?
Places 25 000 queries, then only those below / above size cuts, and prints the sweep time of each pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from apples_amd import synth
from apples_amd.engine import Engine
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
ds = synth.make_dataset(200000, 1000, nq)
nodes = np.array([ds.tree.name_to_node[n] for n in ds.ref_names], np.int32)
eng = Engine(ds.tree, ds.ref_seqs, nodes, method='OLS')


def run(q, tag):
    h, n = eng.upload_queries(q)
    for i in range(2):
        eng.place_resident(h)
    t = eng.timing()
    out = eng.fetch(h, n)
    eng.free_queries(h)
    print(tag, 'queries', n, 'sweep ms', round(t['sweep_ms'], 3), 'per query us', round(t['sweep_ms'] * 1e3 / n, 3),
          'swept nodes mean', round(float(out['n_valid'].mean()), 1), 'dist', round(t['dist_ms'], 2), 'select', round(t['select_ms'], 2), flush=True)
    return out


out = run(ds.query_seqs, 'all')
no = out['n_obs']
print('n_obs percentiles', {p: int(np.percentile(no, p)) for p in (1, 10, 50, 90, 99, 99.9, 100)}, 'over 4096:', int((no > 4096).sum()))
print('n_valid percentiles', {p: int(np.percentile(out['n_valid'], p)) for p in (50, 90, 99, 99.9, 100)})
order = np.argsort(no)
for frac in (0.5, 0.9, 0.99):
    k = int(len(order) * frac)
    run(ds.query_seqs[np.sort(order[:k])], 'smallest %.0f %%' % (frac * 100))
run(ds.query_seqs[np.sort(order[int(len(order) * 0.99):])], 'largest 1 %')
run(ds.query_seqs[np.sort(order[int(len(order) * 0.9):])], 'largest 10 %')
