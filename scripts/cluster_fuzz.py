"""Differential fuzz of the clustered fast path (the command line's default route): random backbone sizes, alignment
lengths, gap rates, cluster diameters (from singletons and pairs to clusters of hundreds), thresholds, -b values,
methods and device batch sizes; placements must be byte-identical between
  the default (cluster-major member distances, phase 4 for the listed queries, vector loads in k_select),
  APPLES_CLUSTER_BY_QUERY=1 (a thread per (query, member) pair),
  APPLES_NO_CLUSTER_TOPUP=1 (the listed queries through full rows + k_select), and
  APPLES_NO_FUSE=1 (full rows + general selection for every query).
The default route is also checked against the C oracle where that is cheap (at most 1 500 leaves).
usage: cluster_fuzz.py [seed] [configurations]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncfg = int(sys.argv[2]) if len(sys.argv) > 2 else 24
code = ("import sys, os, hashlib, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'oracle'))\n"
        "from oracle_c import COracle\n"
        "from apples_amd import synth, treecluster\n"
        "from apples_amd.engine import Engine, jc69_lut\n"
        "from apples_amd.fasta import Alignment\n"
        "from apples_amd.reference import ReducedReference\n"
        "rng = np.random.default_rng(%d)\n"
        "for c in range(%d):\n"
        "    n = int(rng.choice([60, 257, 600, 1500, 5000, 12000])); L = int(rng.integers(40, 2047)); nq = int(rng.integers(1, 900))\n"
        "    gap = float(rng.choice([0.0, 0.05, 0.3, 0.6])); thr = float(rng.choice([0.0, 0.02, 0.2, 0.5, 1.2])); b = int(rng.choice([3, 25, 200]))\n"
        "    mb = int(rng.choice([0, 0, 32, 96, 160, 512])); m = str(rng.choice(['OLS', 'FM', 'BME', 'BE']))\n"
        "    diam = float(rng.choice([0.01, 0.05, 0.24, 0.4, 0.8]))\n"
        "    d = synth.make_dataset(n, L, nq, gap_rate=gap, seed_tree=200 + c, mean_len=float(rng.choice([0.003, 0.01, 0.05])))\n"
        "    nodes = np.array([d.tree.name_to_node[x] for x in d.ref_names], np.int32)\n"
        "    ref = ReducedReference(Alignment(d.ref_names, d.ref_seqs), False, treecluster.grouped(d.tree, diam))\n"
        "    ca = ref.cluster_arrays()\n"
        "    e = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method=m, threshold=thr, baseobs=b, max_batch=mb)\n"
        "    out = e.place_sequences(d.query_seqs); info = e.describe(); e.close()\n"
        "    orc = '-'\n"
        "    if n <= 1500 and not os.environ.get('APPLES_NO_FUSE') and not os.environ.get('APPLES_CLUSTER_BY_QUERY') and not os.environ.get('APPLES_NO_CLUSTER_TOPUP'):\n"
        "        co = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, method=m, criterion='MLSE', threshold=thr, baseobs=b, lut=jc69_lut(L, 0.001), threads=len(os.sched_getaffinity(0)))\n"
        "        orc = 'oracle-ok' if co.place_sequences(d.query_seqs).tobytes() == out.tobytes() else 'ORACLE-BAD'\n"
        "    print(c, n, L, nq, gap, thr, b, mb, m, diam, info['n_reps'], info['cluster_fused'], int((out['edge'] >= 0).sum()), orc, hashlib.sha1(out.tobytes()).hexdigest()[:16], flush=True)\n"
        % (ROOT, ROOT, seed, ncfg))
envs = ({}, {'APPLES_CLUSTER_BY_QUERY': '1'}, {'APPLES_NO_CLUSTER_TOPUP': '1'}, {'APPLES_NO_FUSE': '1'})
res = []
for env in envs:
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, env=dict(os.environ, **env), timeout=3000)
    if r.returncode != 0:
        print(env, 'FAILED', r.stderr.decode()[-1500:])
        sys.exit(1)
    res.append(r.stdout.decode().strip().splitlines())
bad = 0
for rows in zip(*res):
    hs = [x.split()[-1] for x in rows]
    ok = len(set(hs)) == 1 and 'ORACLE-BAD' not in rows[0]
    bad += not ok
    print('OK ' if ok else 'BAD', rows[0], '|', ' '.join(hs[1:]))
print('mismatches:', bad)
sys.exit(1 if bad else 0)
