"""The distance kernel's HBM roofline point (BASELINE.json's second figure) on a reference that
no cache can hold: T = 1 (`k_jc69`, one query per pass over the packed reference), packed reference
>= 1 GiB (2.8 M rows x L 1000 by default = 1.08 GB of bit planes; the Infinity Cache holds 256 MiB).

    python scripts/hbm_point_probe.py [rows] [queries] > profiles/r02_hbm_point.json
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ... -- python3 scripts/hbm_point_probe.py   (counter pass)

delivered = bytes the kernel moves per query (packed reference once + 8 B per pair out) / HIP-event time;
reported against the 6.3 TB/s the guide measures as achievable and the 8 TB/s spec peak."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from apples_amd.engine import Engine
from apples_amd.tree import parse_newick

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2800000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 48
L = 1000
rng = np.random.default_rng(1)
alpha = np.frombuffer(b'ACGTACGTACGTACGTACG-', np.uint8)  # 5 % gaps
ref = np.empty((rows, L), np.uint8)
for r0 in range(0, rows, 200000):  # (in pieces: the index array of one call would be rows x L bytes too)
    ref[r0:r0 + 200000] = alpha[rng.integers(0, 20, size=(min(200000, rows - r0), L), dtype=np.uint8)]
qry = alpha[rng.integers(0, 20, size=(nq, L), dtype=np.uint8)]
tree = parse_newick('((A:0.1,B:0.2):0.25,(C:0.3,(D:0.2,E:0.2):0.2):0.25);')
eng = Engine(tree, ref, np.full(rows, -1, np.int32), method='OLS', max_batch=nq)
del ref
h, n = eng.upload_queries(qry)
out = {'rows': rows, 'L': L, 'queries': n, 'points': []}
for tile in (1, 4):
    for rep in range(3):
        eng.distances_resident(h, tile)
    ms = eng.timing()['dist_ms']
    info = eng.describe()
    passes = (n + tile - 1) // tile
    moved = passes * info['packed_bytes'] + n * info['n_rows'] * 8.0
    gbs = moved / (ms * 1e-3) / 1e9
    out['points'].append({'query_tile': tile, 'ms': ms, 'us_per_query': ms * 1e3 / n, 'packed_reference_bytes': info['packed_bytes'],
                          'bytes_moved': moved, 'delivered_GBps': gbs, 'frac_of_8000': gbs / 8000.0, 'frac_of_6300_achievable': gbs / 6300.0,
                          'algorithmic_GBps': n * (info['n_rows'] * (L + 8.0) + L) / (ms * 1e-3) / 1e9})
out['device'] = eng.describe()['device']
print(json.dumps(out), flush=True)
eng.close()
