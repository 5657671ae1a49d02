"""One configuration of scripts/cluster_fuzz.py against the C oracle, on the fused route and with APPLES_NO_FUSE=1
(separate processes: the knob is per process).  usage: cluster_fuzz_probe.py seed config [fuse|nofuse]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) < 4:
    for mode in ('fuse', 'nofuse'):
        env = dict(os.environ)
        if mode == 'nofuse':
            env['APPLES_NO_FUSE'] = '1'
        subprocess.run([sys.executable, __file__, sys.argv[1], sys.argv[2], mode], env=env)
    sys.exit(0)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
from oracle_c import COracle
from apples_amd import synth, treecluster
from apples_amd.engine import Engine, jc69_lut
from apples_amd.fasta import Alignment
from apples_amd.reference import ReducedReference
seed, want_c, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(seed)
for c in range(want_c + 1):
    n = int(rng.choice([60, 257, 600, 1500, 5000, 12000])); L = int(rng.integers(40, 2047)); nq = int(rng.integers(1, 900))
    gap = float(rng.choice([0.0, 0.05, 0.3, 0.6])); thr = float(rng.choice([0.0, 0.02, 0.2, 0.5, 1.2])); b = int(rng.choice([3, 25, 200]))
    mb = int(rng.choice([0, 0, 32, 96, 160, 512])); m = str(rng.choice(['OLS', 'FM', 'BME', 'BE']))
    diam = float(rng.choice([0.01, 0.05, 0.24, 0.4, 0.8]))
    mean_len = float(rng.choice([0.003, 0.01, 0.05]))
d = synth.make_dataset(n, L, nq, gap_rate=gap, seed_tree=200 + want_c, mean_len=mean_len)
nodes = np.array([d.tree.name_to_node[x] for x in d.ref_names], np.int32)
ref = ReducedReference(Alignment(d.ref_names, d.ref_seqs), False, treecluster.grouped(d.tree, diam))
ca = ref.cluster_arrays()
e = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method=m, threshold=thr, baseobs=b, max_batch=mb)
got = e.place_sequences(d.query_seqs); e.close()
co = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, method=m, criterion='MLSE', threshold=thr, baseobs=b,
             lut=jc69_lut(L, 0.001), threads=len(os.sched_getaffinity(0)))
want = co.place_sequences(d.query_seqs)
bad = [i for i in range(nq) if got[i].tobytes() != want[i].tobytes()]
if os.environ.get('PROBE_EDGES_ONLY'):
    bad = [i for i in bad if got[i]['edge'] != want[i]['edge']]
print(mode, 'config', want_c, (n, L, nq, gap, thr, b, mb, m, diam), 'differing queries:', len(bad))
for i in bad[:6]:
    print('   q', i, 'got', got[i], 'want', want[i])
