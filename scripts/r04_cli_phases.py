"""Phase log of the command line at C3 size (run_apples.py --debug: its own INFO lines), singleton clusters and default clusters."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from apples_amd import synth
n_leaves, L, nq = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (200000, 1000, 100000)
d = synth.make_dataset(n_leaves, L, nq)
tmp = tempfile.mkdtemp()
def wf(path, names, seqs):
    with open(path, 'w') as f:
        for n, s in zip(names, seqs):
            f.write('>%s\n%s\n' % (n, s.tobytes().decode()))
open(os.path.join(tmp, 'tree.nwk'), 'w').write(d.newick + '\n')
wf(os.path.join(tmp, 'ref.fa'), d.ref_names, d.ref_seqs)
wf(os.path.join(tmp, 'query.fa'), d.query_names, d.query_seqs)
for extra in (['--no-clusters'], ['--no-clusters'], []):
    t = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'run_apples.py'), '-s', os.path.join(tmp, 'ref.fa'), '-q',
                        os.path.join(tmp, 'query.fa'), '-t', os.path.join(tmp, 'tree.nwk'), '-o', os.path.join(tmp, 'o.jplace'), '-D', '--debug'] + extra,
                       capture_output=True, text=True)
    print(extra, 'wall %.2f s' % (time.time() - t))
    for l in r.stderr.strip().splitlines():
        if 'seconds' in l or 'phase' in l:
            print('   ', l[-150:])
