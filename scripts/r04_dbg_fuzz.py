import os, sys
import numpy as np
ROOT='/root/repo' if os.path.exists('/root/repo/oracle') else os.environ.get('GRAFT_REPO_ROOT','.')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'oracle'))
from oracle_c import COracle
from apples_amd import synth
from apples_amd.engine import Engine
rng = np.random.default_rng(11)
for c in range(27):
    n = int(rng.choice([40, 257, 600, 1500, 5000, 20000])); L = int(rng.integers(7, 1200)); nq = int(rng.integers(1, 700))
    gap = float(rng.choice([0.0, 0.05, 0.3, 0.6])); thr = float(rng.choice([0.03, 0.1, 0.2, 0.24, 0.5])); b = int(rng.choice([3, 25, 200]))
    mb = int(rng.choice([0, 0, 32, 96, 160, 512])); m = str(rng.choice(['OLS', 'FM', 'BME', 'BE']))
    ml = float(rng.choice([0.003, 0.01, 0.05]))
    if c < 26: continue
    d = synth.make_dataset(n, L, nq, protein=True, gap_rate=gap, seed_tree=400 + c, mean_len=ml)
    q = d.query_seqs.copy()
    if nq > 6:
        q[3] = d.ref_seqs[11 % n]; q[4] = ord('-'); q[5, ::2] = ord('x')
    nodes = np.array([d.tree.name_to_node[x] for x in d.ref_names], np.int32)
    e = Engine(d.tree, d.ref_seqs, nodes, protein=True, method=m, threshold=thr, baseobs=b, max_batch=mb)
    got = e.place_sequences(q); e.close()
    co = COracle(d.tree, d.ref_seqs, nodes, protein=True, method=m, criterion='MLSE', threshold=thr, baseobs=b, threads=8)
    want = co.place_sequences(q)
    bad = np.nonzero(got['edge'] != want['edge'])[0]
    print(n, L, nq, gap, thr, b, m, 'bad', bad)
    for i in bad[:6]:
        print(i, got[i]); print('   ', want[i])
