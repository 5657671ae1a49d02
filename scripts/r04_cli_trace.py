"""In-process timing of the command line's steps at C3 size (wrappers around the calls, no profiler)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
del d
import run_apples
from apples_amd import engine, worker, fasta, tree, jplace, reference
T0 = time.time()
def wrap(mod, name, label=None):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.time()
        try:
            return f(*a, **k)
        finally:
            print('%-34s %.3f s  (at %.3f)' % (label or name, time.time() - t, time.time() - T0), flush=True)
    setattr(mod, name, g)
wrap(engine.Engine, '__init__', 'Engine.__init__')
wrap(engine.Engine, 'place_sequences')
wrap(engine.Engine, 'close', 'Engine.close')
wrap(worker.QueryWorker, '_rows')
wrap(run_apples, 'read_alignment')
wrap(run_apples, 'read_tree')
wrap(run_apples, 'extended_newick')
wrap(run_apples, 'write_native')
wrap(run_apples, 'ReducedReference')
T0 = time.time()
run_apples.main(['-s', os.path.join(tmp, 'ref.fa'), '-q', os.path.join(tmp, 'query.fa'), '-t', os.path.join(tmp, 'tree.nwk'), '-o',
                 os.path.join(tmp, 'o.jplace'), '-D', '--no-clusters'])
print('main %.3f s' % (time.time() - T0))
