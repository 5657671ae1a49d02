#!/usr/bin/env python3
"""The legs of config 3's size in the shapes real inputs have (bench.py: variant_dataset / other_shapes), one by one, each
inside its own try: what `python bench.py` adds to its line as `other_shapes`, without the rest of the line.

    python scripts/shape_legs.py [leg ...] > gpurun_out/shape_legs.json
"""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from apples_amd import synth  # noqa: E402

want = sys.argv[1:]
n_leaves, L, Q, protein, method, thr = bench.WORKLOADS['c3']
ds = synth.make_dataset(n_leaves, L, Q)
res = {}


def leg(key, fn):
    if want and key not in want:
        return
    t0 = time.time()
    try:
        res[key] = fn()
    except Exception as e:  # noqa: BLE001
        res[key] = {'error': repr(e), 'trace': traceback.format_exc()[-1500:]}
    res[key]['leg_wall_s'] = time.time() - t0
    print(key, json.dumps(res[key])[:600], file=sys.stderr, flush=True)


leg('c3', lambda: bench.other_workload('c3', 0, ds=ds))
leg('c3-clustered', lambda: bench.other_workload('c3-clustered', 0, ds=ds))
for v in ('unrooted', 'polytomies', 'deep', 'dots', 'L4000'):
    keys = ['c3-' + v] + (['c3-%s-clustered' % v] if v in ('unrooted', 'polytomies', 'deep') else [])
    if want and not any(k in want for k in keys):
        continue
    q = 25000 if v == 'L4000' else 0
    dv = bench.variant_dataset(ds, v, n_leaves, L, q or Q, protein)
    leg('c3-' + v, lambda: bench.other_workload('c3', 0, ds=dv, variant=v, prepared=True, queries=q))
    if len(keys) > 1:
        leg(keys[1], lambda: bench.other_workload('c3-clustered', 0, ds=dv, variant=v, prepared=True))
    del dv
leg('c3-clustered-300k', lambda: bench.other_workload('c3-clustered', 0, variant='300k'))
print(json.dumps(res))
