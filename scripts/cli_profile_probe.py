"""Where the command line spends its time at benchmark size: scripts/cli_scale_probe.py's files,
run_apples.py under cProfile, top of the cumulative list."""
import os, pstats, subprocess, sys, tempfile
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
prof = os.path.join(tmp, 'prof.out')
r = subprocess.run([sys.executable, '-m', 'cProfile', '-o', prof, os.path.join(ROOT, 'run_apples.py'), '-s',
                    os.path.join(tmp, 'ref.fa'), '-q', os.path.join(tmp, 'query.fa'), '-t', os.path.join(tmp, 'tree.nwk'),
                    '-o', os.path.join(tmp, 'out.jplace'), '-D'] + sys.argv[4:], capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-2000:]
pstats.Stats(prof).sort_stats('cumulative').print_stats(28)
