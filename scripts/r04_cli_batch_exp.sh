# the command line at C3 size with different budgets for the device batch buffers, runs back to back (no pause between them):
# what the driver's scrubbing of the previous process's buffers costs the next one
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, subprocess, sys, tempfile, time
sys.path.insert(0, '.')
from apples_amd import synth
d = synth.make_dataset(200000, 1000, 100000)
tmp = tempfile.mkdtemp()
def wf(path, names, seqs):
    with open(path, 'w') as f:
        for n, s in zip(names, seqs):
            f.write('>%s\n%s\n' % (n, s.tobytes().decode()))
open(os.path.join(tmp, 'tree.nwk'), 'w').write(d.newick + '\n')
wf(os.path.join(tmp, 'ref.fa'), d.ref_names, d.ref_seqs)
wf(os.path.join(tmp, 'query.fa'), d.query_names, d.query_seqs)
for gib in ('12', '12', '24', '24', '48', '48', '', ''):
    env = dict(os.environ)
    if gib:
        env['APPLES_BATCH_GIB'] = gib
    t = time.time()
    r = subprocess.run([sys.executable, 'run_apples.py', '-s', os.path.join(tmp, 'ref.fa'), '-q', os.path.join(tmp, 'query.fa'), '-t',
                        os.path.join(tmp, 'tree.nwk'), '-o', os.path.join(tmp, 'o.jplace'), '-D', '--debug', '--no-clusters'], capture_output=True, text=True, env=env)
    ph = [l.split('in ')[-1].split(' ')[0] for l in r.stderr.splitlines() if 'seconds' in l]
    print('APPLES_BATCH_GIB=%-3s wall %.2f s  phases (tree, reference, queries, all) %s' % (gib or '-', time.time() - t, ph), flush=True)
PY
