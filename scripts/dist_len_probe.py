"""Distance-phase time vs alignment length, matrix-core kernel vs bit-plane kernel (run twice with/without APPLES_NO_DIST_MFMA)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from apples_amd import synth
from apples_amd.engine import Engine
for L in (1000, 4000):
    ds = synth.make_dataset(10000, L, 4096)
    nodes = np.array([ds.tree.name_to_node[n] for n in ds.ref_names], np.int32)
    eng = Engine(ds.tree, ds.ref_seqs, nodes, method='OLS')
    h, n = eng.upload_queries(ds.query_seqs)
    for i in range(3):
        eng.place_resident(h)
    t = eng.timing()
    print('L', L, 'mfma' if not os.environ.get('APPLES_NO_DIST_MFMA') else 'valu', {k: round(float(v), 3) for k, v in t.items() if k in ('dist_ms', 'select_ms', 'sweep_ms')}, flush=True)
    eng.close()
