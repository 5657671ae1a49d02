#!/usr/bin/env python3
"""Would the clustered route gain from two device batches in flight (distance + selection of one beside the sweep of the other)?
The cheapest way to ask the device: two contexts on the one GPU, each with its own streams and workspace, each placing half of
config 3's queries from its own host thread, against one context placing them all.  No library change.

    python scripts/r06_two_contexts_exp.py [c3|c3-clustered] > gpurun_out/r06_two_contexts_exp.txt
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from apples_amd import synth  # noqa: E402
from apples_amd.engine import Engine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'c3-clustered'
n_leaves, L, Q, protein, method, thr = bench.WORKLOADS[name]
ds = synth.make_dataset(n_leaves, L, Q, protein=protein)
nodes = np.array([ds.tree.name_to_node[n] for n in ds.ref_names], np.int32)
clusters = bench.make_clusters(ds, thr, protein) if name.endswith('-clustered') else None


def engine(batch_gib=0):
    kw = {'batch_gib': batch_gib} if batch_gib else {}
    return Engine(ds.tree, ds.ref_seqs, nodes, clusters=clusters, protein=protein, method=method, criterion='MLSE', threshold=thr,
                  baseobs=25, overlap=0.001, device=0, **kw)


qall = np.ascontiguousarray(ds.query_seqs[:Q])


def timed(fn, reps=5):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append((time.perf_counter() - t0) * 1e3)
    return min(t), float(np.median(t))


one = engine()
ref = one.place_sequences(qall)
print('%s: one context, %d queries: min %.2f ms, median %.2f ms  (batch %s)' % ((name, Q) + timed(lambda: one.place_sequences(qall)) +
                                                                                 (one.describe().get('batch'),)), flush=True)
one.close()  # (its workspace: half of the device's memory)
for parts in (2, 3):
    # (each context sizes its batch from the memory that is free when it first runs: a fixed budget keeps them equal)
    engs = [engine(batch_gib=64 if parts == 2 else 44) for _ in range(parts)]
    cut = [Q * k // parts for k in range(parts + 1)]
    qs = [np.ascontiguousarray(qall[cut[k]:cut[k + 1]]) for k in range(parts)]
    outs = [None] * parts

    def run_one(k):
        outs[k] = engs[k].place_sequences(qs[k])

    def together():
        th = [threading.Thread(target=run_one, args=(k,)) for k in range(parts)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    def in_turn():
        for k in range(parts):
            run_one(k)

    a = timed(in_turn)
    b = timed(together)
    same = all(np.array_equal(outs[k][f], ref[f][cut[k]:cut[k + 1]]) for k in range(parts) for f in ('edge', 'n_valid'))
    print('%d contexts, %d queries each: one after the other min %.2f ms (median %.2f); side by side min %.2f ms (median %.2f); '
          'placements equal the one context\'s: %s; batch %s' % ((parts, Q // parts) + a + b + (same, engs[0].describe().get('batch'))), flush=True)
    del engs
