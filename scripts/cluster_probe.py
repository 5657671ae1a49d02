"""Clustered-reference mode (the CLI default) at C2 size: kernel split vs the all-singleton mode."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from apples_amd import synth, treecluster
from apples_amd.engine import Engine
from apples_amd.fasta import Alignment
from apples_amd.reference import ReducedReference
n, L, Q = 10000, 1000, 10000
ds = synth.make_dataset(n, L, Q)
aln = Alignment(ds.ref_names, ds.ref_seqs)
nodes = np.array([ds.tree.name_to_node[x] for x in ds.ref_names], np.int32)
t = time.time(); cl = treecluster.grouped(ds.tree, 0.24); rr = ReducedReference(aln, False, cl); print('clustering+consensus s', time.time() - t, 'clusters', len(cl))
for label, clusters in (('singleton', None), ('clustered', rr.cluster_arrays())):
    eng = Engine(ds.tree, ds.ref_seqs, nodes, clusters=clusters, method='OLS')
    h, nq = eng.upload_queries(ds.query_seqs)
    for i in range(3):
        t = time.time(); eng.place_resident(h); dt = time.time() - t
    out = eng.fetch(h, nq)
    print(label, round(dt * 1e3, 2), 'ms', {k: round(float(v), 2) for k, v in eng.timing().items()}, 'mean n_obs', out['n_obs'].mean(), eng.describe()['n_reps'])
    eng.close()
