"""End-to-end command line at benchmark size: writes synthetic FASTA/Newick files, runs run_apples.py
as a subprocess (alignment input, all-singleton clusters, backbone as given) and reports wall time,
the CLI's own phase log and a sanity check of the jplace."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from apples_amd import synth

n_leaves, L, nq = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (10000, 1000, 10000)
d = synth.make_dataset(n_leaves, L, nq)
tmp = tempfile.mkdtemp()
def wf(path, names, seqs):
    with open(path, 'w') as f:
        for n, s in zip(names, seqs):
            f.write('>%s\n%s\n' % (n, s.tobytes().decode()))
open(os.path.join(tmp, 'tree.nwk'), 'w').write(d.newick + '\n')
wf(os.path.join(tmp, 'ref.fa'), d.ref_names, d.ref_seqs)
wf(os.path.join(tmp, 'query.fa'), d.query_names, d.query_seqs)
out = os.path.join(tmp, 'out.jplace')
# every configuration three times, a pause before each: a process that starts right after another one has exited waits for the
# driver to scrub that one's device memory (seconds for tens of GiB; scripts/r04_cli_batch_exp.sh), which says nothing about this one
for clusters in (['--no-clusters'], []):
    walls = []
    for rep in range(3):
        time.sleep(6)
        t = time.time()
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'run_apples.py'), '-s', os.path.join(tmp, 'ref.fa'), '-q',
                            os.path.join(tmp, 'query.fa'), '-t', os.path.join(tmp, 'tree.nwk'), '-o', out, '-D', '--debug'] + clusters,
                           capture_output=True, text=True)
        walls.append(time.time() - t)
        assert r.returncode == 0, r.stderr[-2000:]
        phases = [l.split('] ', 1)[-1] for l in r.stderr.strip().splitlines() if 'seconds' in l]
        print('  run %d: wall %.2f s | %s' % (rep, walls[-1], ' | '.join(phases)[:400]), flush=True)
    j = json.load(open(out))
    print('clusters' if not clusters else 'no clusters', 'wall best of 3 %.2f s (%s)' % (min(walls), ', '.join('%.2f' % w for w in walls)),
          'placements', len(j['placements']), 'bytes', os.path.getsize(out), flush=True)
