"""Per-pass kernel split with the command line's default reduced reference (clusters at 1.2 x -f)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from apples_amd import synth, treecluster
from apples_amd.engine import Engine
from apples_amd.fasta import Alignment
from apples_amd.reference import ReducedReference
n_leaves, L, nq = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (10000, 1000, 10000)
d = synth.make_dataset(n_leaves, L, nq)
nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
ref = ReducedReference(Alignment(d.ref_names, d.ref_seqs), False, treecluster.grouped(d.tree, 0.24))
ca = ref.cluster_arrays()
print('clusters with consensus rows:', len(ca[0]), 'representatives:', len(ca[1]))
for label, kw in (('clustered', dict(clusters=ca)), ('singletons', {})):
    eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS', **kw)
    h, n = eng.upload_queries(d.query_seqs)
    for i in range(3):
        t = time.time(); eng.place_resident(h); dt = time.time() - t
    out = eng.fetch(h, n)
    print(label, 'ms', round(dt * 1e3, 3), {k: round(float(v), 3) for k, v in eng.timing().items() if k.endswith('_ms')},
          'mean n_obs', out['n_obs'].mean(), 'q/s', round(n / dt))
    eng.close()
