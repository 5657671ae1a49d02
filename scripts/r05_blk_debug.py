#!/usr/bin/env python3
"""clade blocks: one fuzz configuration, default against no_blocks, row by row"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from apples_amd import synth, treecluster
from apples_amd.engine import Engine
from apples_amd.fasta import Alignment
from apples_amd.reference import ReducedReference

n, L, nq, gap, thr, b, mb, m, crit, neg, diam, seed_tree, mean_len = 12000, 889, 246, 0.6, 0.02, 25, int(os.environ.get('MB', 96)), 'BME', 'ME', False, 0.01, 201, None
rng = np.random.default_rng(1)
# replay the stream of tests/test_gpu_fuzz.py::test_clustered_routes_agree[1] up to configuration CFG
CFG = int(os.environ.get('CFG', 1))
for c in range(CFG + 1):
    n = int(rng.choice([60, 257, 600, 1500, 5000, 12000])); L = int(rng.integers(40, 2047)); nq = int(rng.integers(1, 900))
    gap = float(rng.choice([0.0, 0.05, 0.3, 0.6])); thr = float(rng.choice([0.0, 0.02, 0.2, 0.5, 1.2])); b = int(rng.choice([3, 25, 200]))
    mb = int(rng.choice([0, 0, 32, 96, 160, 512])); m = str(rng.choice(['OLS', 'FM', 'BME', 'BE']))
    diam = float(rng.choice([0.01, 0.05, 0.24, 0.4, 0.8]))
    mean_len = float(rng.choice([0.003, 0.01, 0.05]))
r = np.random.default_rng([1, CFG, 77]); crit = str(r.choice(('MLSE', 'ME', 'HYBRID'))); neg = bool(r.integers(0, 2))
print('cfg', CFG, n, L, nq, gap, thr, b, mb, m, crit, neg, diam, mean_len)
d = synth.make_dataset(n, L, nq, gap_rate=gap, seed_tree=200 + CFG, mean_len=mean_len)
nodes = np.array([d.tree.name_to_node[x] for x in d.ref_names], np.int32)
ca = ReducedReference(Alignment(d.ref_names, d.ref_seqs), False, treecluster.grouped(d.tree, diam)).cluster_arrays()
out = {}
for name, dbg in (('default', ()), ('no_blocks', ('no_blocks',))):
    e = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method=m, criterion=crit, negative=neg, threshold=thr, baseobs=b, max_batch=mb, debug=dbg)
    out[name] = e.place_sequences(d.query_seqs)
    out[name + '2'] = e.place_sequences(d.query_seqs)
    print(name, {k: v for k, v in e.describe().items() if k in ('cluster_blocks', 'n_reps', 'batch', 'sweep_layout', 'cluster_fused')})
    e.close()
a, bb = out['default'], out['no_blocks']
bad = np.nonzero([x.tobytes() != y.tobytes() for x, y in zip(a, bb)])[0]
print('differ', len(bad), 'of', nq, '; default repeat differs', sum(x.tobytes() != y.tobytes() for x, y in zip(a, out['default2'])))
for i in bad[:12]:
    print(i, a[i], bb[i])
