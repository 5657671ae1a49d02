"""C3-size sweep with the level lists built by merging (APPLES_SWEEP_MERGE=1) against the tagged node map.
Each configuration runs in its own process; placements must be byte-identical."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from apples_amd import synth\n"
        "from apples_amd.engine import Engine\n"
        "ds = synth.make_dataset(200000, 1000, %d)\n"
        "nodes = np.array([ds.tree.name_to_node[n] for n in ds.ref_names], np.int32)\n"
        "for m in ('OLS', 'BME'):\n"
        "    eng = Engine(ds.tree, ds.ref_seqs, nodes, method=m)\n"
        "    h, n = eng.upload_queries(ds.query_seqs)\n"
        "    for i in range(3): eng.place_resident(h)\n"
        "    t = eng.timing(); out = eng.fetch(h, n); eng.close()\n"
        "    sys.stderr.write('%%s sweep ms %%.3f select %%.3f dist %%.3f\\n' %% (m, t['sweep_ms'], t['select_ms'], t['dist_ms']))\n"
        "    sys.stdout.buffer.write(out.tobytes())\n" % (ROOT, nq))
ref = None
for env in ({}, {'APPLES_SWEEP_MERGE': '1'}):
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, env=dict(os.environ, **env), timeout=1500)
    same = None if ref is None else r.stdout == ref
    if ref is None:
        ref = r.stdout
    print(env, 'rc', r.returncode, ' | '.join(r.stderr.decode().strip().splitlines()[-2:])[:300], 'identical to default:', same, flush=True)
    if same is False:
        import numpy as np
        a = np.frombuffer(ref, np.uint8).reshape(-1, 40); b = np.frombuffer(r.stdout, np.uint8).reshape(-1, 40)
        bad = np.nonzero((a != b).any(1))[0]
        print('differing rows', len(bad), 'of', len(a), bad[:10])
