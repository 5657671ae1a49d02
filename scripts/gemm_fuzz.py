"""Differential fuzz of the GEMM-form fused distance pass: random reference sizes, alignment lengths, gap rates,
thresholds, -b values, methods and device batch sizes; placements must be byte-identical with APPLES_NO_DIST_GEMM=1
(bit-plane-fed matrix-core kernel; with the sweep's merged level lists forced) and APPLES_NO_FUSE=1 (full rows; with the
sweep's node map forced); the default run uses the node bits in LDS at these tree sizes."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncfg = int(sys.argv[2]) if len(sys.argv) > 2 else 24
code = ("import sys, hashlib, numpy as np; sys.path.insert(0, %r)\n"
        "from apples_amd import synth\n"
        "from apples_amd.engine import Engine\n"
        "rng = np.random.default_rng(%d)\n"
        "for c in range(%d):\n"
        "    n = int(rng.choice([40, 257, 600, 1500, 5000, 20000])); L = int(rng.integers(20, 2047)); nq = int(rng.integers(1, 700))\n"
        "    gap = float(rng.choice([0.0, 0.05, 0.3, 0.6])); thr = float(rng.choice([0.05, 0.2, 0.5, 1.2])); b = int(rng.choice([3, 25, 200]))\n"
        "    mb = int(rng.choice([0, 0, 32, 96, 160, 512])); m = str(rng.choice(['OLS', 'FM', 'BME', 'BE']))\n"
        "    d = synth.make_dataset(n, L, nq, gap_rate=gap, seed_tree=100 + c, mean_len=float(rng.choice([0.003, 0.01, 0.05])))\n"
        "    nodes = np.array([d.tree.name_to_node[x] for x in d.ref_names], np.int32)\n"
        "    e = Engine(d.tree, d.ref_seqs, nodes, method=m, threshold=thr, baseobs=b, max_batch=mb)\n"
        "    out = e.place_sequences(d.query_seqs); info = e.describe(); e.close()\n"
        "    print(c, n, L, nq, gap, thr, b, mb, m, info['fused_distance_pass'], int((out['edge'] >= 0).sum()), hashlib.sha1(out.tobytes()).hexdigest()[:16], flush=True)\n"
        % (ROOT, seed, ncfg))
res = []
for env in ({}, {'APPLES_NO_DIST_GEMM': '1', 'APPLES_SWEEP_MERGE': '1'}, {'APPLES_NO_FUSE': '1', 'APPLES_NODE_MAP': '1'}):
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, env=dict(os.environ, **env), timeout=3000)
    if r.returncode != 0:
        print(env, 'FAILED', r.stderr.decode()[-1500:])
        sys.exit(1)
    res.append(r.stdout.decode().strip().splitlines())
bad = 0
for a, b, c in zip(*res):
    ha, hb, hc = a.split()[-1], b.split()[-1], c.split()[-1]
    ok = ha == hb == hc
    bad += not ok
    print('OK ' if ok else 'BAD', a, '|', b.split()[-3], '|', c.split()[-3])
print('mismatches:', bad)
sys.exit(1 if bad else 0)
