"""Differential fuzz of the -d (distance table) route against the C oracle: random backbone sizes (odd and even numbers
of columns: rows then start 8- or 16-byte aligned, which picks the loads of the selection kernels), columns in random
order, some columns not in the tree, negative and zero entries, duplicated small values, thresholds, -b, methods; the
default route (streaming selection), APPLES_NO_STREAM_SELECT=1 (general selection) and APPLES_NO_TOPUP_KERNEL both must be
byte-identical with the oracle.  usage: table_fuzz.py [seed] [configurations]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncfg = int(sys.argv[2]) if len(sys.argv) > 2 else 24
code = ("import sys, os, hashlib, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'oracle'))\n"
        "from oracle_c import COracle\n"
        "from apples_amd import synth\n"
        "from apples_amd.engine import Engine\n"
        "rng = np.random.default_rng(%d)\n"
        "for c in range(%d):\n"
        "    n = int(rng.choice([33, 64, 257, 1000, 4097, 20001])); nq = int(rng.integers(1, 200))\n"
        "    thr = float(rng.choice([0.0, 0.05, 0.2, 1.0])); b = int(rng.choice([3, 25, 200])); m = str(rng.choice(['OLS', 'FM', 'BME', 'BE']))\n"
        "    d = synth.make_dataset(n, 8, nq, seed_tree=300 + c, mean_len=float(rng.choice([0.003, 0.01, 0.05])))\n"
        "    nodes = np.array([d.tree.name_to_node[x] for x in d.ref_names], np.int32)\n"
        "    D = synth.noisy_distance_rows(d.tree, d.query_leaf, d.query_pendant, list(range(nq)), seed_noise=c)\n"
        "    perm = rng.permutation(n); D = np.ascontiguousarray(D[:, perm]); cols = nodes[perm].copy()\n"
        "    off = rng.random(n) < float(rng.choice([0.0, 0.02, 0.3])); cols[off] = -1              # columns that are not tree leaves\n"
        "    neg = rng.random(D.shape) < float(rng.choice([0.0, 0.01])); D[neg] = -1.0              # invalid entries\n"
        "    zer = rng.random(D.shape) < float(rng.choice([0.0, 0.0005])); D[zer] = 0.0             # exact matches (also in columns outside the tree)\n"
        "    tie = rng.random(D.shape) < 0.01; D[tie] = np.round(D[tie], 2)                          # ties\n"
        "    co = COracle(d.tree, method=m, criterion='MLSE', threshold=thr, baseobs=b, threads=len(os.sched_getaffinity(0)))\n"
        "    want = co.place_distances(D, cols)\n"
        "    e = Engine(d.tree, None, method=m, criterion='MLSE', threshold=thr, baseobs=b)\n"
        "    got = e.place_distances(D, cols); e.close()\n"
        "    print(c, n, nq, thr, b, m, int(off.sum()), int((got['edge'] >= 0).sum()), 'OK' if got.tobytes() == want.tobytes() else 'BAD', flush=True)\n"
        % (ROOT, ROOT, seed, ncfg))
bad = 0
for env in ({}, {'APPLES_NO_STREAM_SELECT': '1'}, {'APPLES_NO_TOPUP_KERNEL': '1'}):
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, env=dict(os.environ, **env), timeout=3000)
    if r.returncode != 0:
        print(env, 'FAILED', r.stderr.decode()[-1500:])
        sys.exit(1)
    rows = r.stdout.decode().strip().splitlines()
    nb = sum(1 for x in rows if x.endswith('BAD'))
    bad += nb
    print(env, len(rows), 'configurations,', nb, 'differ from the C oracle')
    for x in rows:
        if x.endswith('BAD'):
            print('   ', x)
print('mismatches:', bad)
sys.exit(1 if bad else 0)
