"""scoredist routes against one another on the GPU: matrix-core filter + exact candidates (default), every pair with the
early exit (no_sd_gemm), full rows + general selection (no_fuse); same bytes expected."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apples_amd import synth
from apples_amd.engine import Engine


def routes(N, L, Q, thr=0.2, b=25, seed=3, max_batch=0):
    d = synth.make_dataset(N, L, Q, protein=True, seed_query=seed)
    q = d.query_seqs.copy()
    if Q > 5:
        q[3] = d.ref_seqs[11 % N]
        q[4] = ord('-')
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    outs = {}
    for name, dbg in (('gemm', ()), ('gemm_fp6', ('sd_fp6',)), ('early', ('no_sd_gemm',)), ('nofuse', ('no_fuse',))):
        e = Engine(d.tree, d.ref_seqs, nodes, protein=True, method='FM', threshold=thr, baseobs=b, debug=dbg, max_batch=max_batch)
        t0 = time.perf_counter()
        outs[name] = e.place_sequences(q)
        dt = time.perf_counter() - t0
        tm = e.timing()
        e.close()
        print('  %-7s %.1f ms  dist %.2f select %.2f sweep %.2f' % (name, dt * 1e3, tm['dist_ms'], tm['select_ms'], tm['sweep_ms']), flush=True)
    ok = all(outs[k].tobytes() == outs['nofuse'].tobytes() for k in ('gemm', 'gemm_fp6', 'early'))
    if not ok:
        for k in ('gemm', 'gemm_fp6', 'early'):
            bad = np.nonzero([a.tobytes() != b_.tobytes() for a, b_ in zip(outs[k], outs['nofuse'])])[0]
            print('  MISMATCH', k, len(bad), bad[:10])
            for i in bad[:3]:
                print('   ', outs[k][i], outs['nofuse'][i])
    print('N %d L %d Q %d thr %g b %d batch %d: %s' % (N, L, Q, thr, b, max_batch, 'OK' if ok else 'FAIL'), flush=True)
    return ok


if __name__ == '__main__':
    good = True
    for args in ((600, 100, 70), (5000, 300, 900), (5000, 300, 900, 0.03, 60), (3000, 7, 300), (2500, 513, 257, 0.2, 25, 3, 96),
                 (20000, 500, 3000, 0.2, 25, 5), (700, 40, 33, 0.24, 5)):
        good &= routes(*args)
    print('ALL OK' if good else 'SOME FAILED')
    sys.exit(0 if good else 1)
