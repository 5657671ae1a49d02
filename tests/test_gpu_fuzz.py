"""Differential fuzz of the routes through the hot path, inside `pytest -m gpu` (round 3's scripts/*_fuzz.py, which found
the one real selection bug of that round -- a query identical to two members of one cluster, `k_select`'s first-zero
bookkeeping -- lived outside the suite).  Every test draws random configurations from a fixed seed (backbone size,
alignment length, gap rate, threshold, -b, method, device batch size, ...) and requires byte-identical placements between
routes that share no distance, selection or sweep code, crossed inside one process through `apples_params.debug`; the
small backbones also against the C oracle (what the reference computes: apples/Reference.py:138-154,
apples/PoolQueryWorker.py:63-98).  Every configuration also draws the selection criterion (-c MLSE / ME / HYBRID,
apples/Algorithm.py:76-101) and -n (apples/util.py:32-50).  Two seeds x 40 configurations per family, five families."""
import os
import sys

import numpy as np
import pytest

from helpers import ROOT

sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from oracle_c import COracle  # noqa: E402

from apples_amd import synth, treecluster  # noqa: E402
from apples_amd.engine import Engine, jc69_lut  # noqa: E402
from apples_amd.fasta import Alignment  # noqa: E402
from apples_amd.reference import ReducedReference  # noqa: E402

pytestmark = pytest.mark.gpu
NTHREADS = len(os.sched_getaffinity(0))
NCFG = 40
# APPLES_FUZZ_SEEDS="20-59": extra seeds for every family (a fuzz campaign outside the suite's two fixed ones)
_extra = os.environ.get('APPLES_FUZZ_SEEDS', '')
EXTRA_SEEDS = list(range(int(_extra.split('-')[0]), int(_extra.split('-')[1]) + 1)) if '-' in _extra else []


def _place(routes, make, queries, place='place_sequences'):
    """placements per route: make(debug) -> Engine"""
    out = {}
    only = os.environ.get('APPLES_FUZZ_ROUTES')  # (chasing a crash: "default,ragged_dry")
    for name, dbg in routes:
        if only and name not in only.split(','):
            continue
        if os.environ.get('APPLES_FUZZ_TRACE'):  # (a campaign's crash: which route of which configuration)
            print('route', name, file=sys.stderr, flush=True)
        if isinstance(dbg, dict):  # (a route by knobs of the context: make(debug, knobs))
            e = make((), dbg)
        else:
            e = make(dbg)
        out[name] = getattr(e, place)(*queries) if isinstance(queries, tuple) else getattr(e, place)(queries)
        e.close()
    return out


CRITERIA = ('MLSE', 'ME', 'HYBRID')


def _crit(seed, c):
    """(-c, -n) of configuration c: apples/Algorithm.py:76-101, apples/util.py:32-50.  Drawn apart from the configuration
    stream (seed 1 of the clustered family is the stream that found round 3's fault: it stays what it was)."""
    r = np.random.default_rng([seed, c, 77])
    return str(r.choice(CRITERIA)), bool(r.integers(0, 2))


def _shaped(seed, c, n, make):
    """The backbone's shape for configuration c, drawn apart from the configuration stream: the strictly binary random-join tree
    (a third of the configurations), the same tree unrooted (a root trifurcation), with a share of its internal nodes dissolved
    into their parents (polytomies of any degree, apples/OLS.py:36,59), hung on a caterpillar spine (more than 254 levels from
    600 leaves on: the window of the lean sweep's per-level offsets), or both.  make(spine) -> dataset; returns (dataset, tag)."""
    r = np.random.default_rng([seed, c, 78])
    kind = str(r.choice(['binary', 'binary', 'unrooted', 'polytomies', 'deep', 'deep+polytomies']))
    frac = float(r.choice([0.02, 0.1, 0.4]))
    d = make(int(min(290, n // 2)) if kind.startswith('deep') else 0)
    if kind == 'unrooted':
        d.tree = synth.reshape_tree(d.tree, 'unrooted')
    elif kind.endswith('polytomies'):
        d.tree = synth.reshape_tree(d.tree, 'polytomies', seed=1000 * seed + c, frac=frac)
    return d, ' shape %s%s' % (kind, ' %g' % frac if kind.endswith('polytomies') else '')


def _diff(a, b):
    bad = np.nonzero([x.tobytes() != y.tobytes() for x, y in zip(a, b)])[0]
    return '%d rows differ, first %s: %s / %s' % (len(bad), bad[:5], a[bad[0]] if len(bad) else '', b[bad[0]] if len(bad) else '')


@pytest.mark.parametrize('seed', [1, 9] + EXTRA_SEEDS)
def test_clustered_routes_agree(seed):
    """The command line's default route (clustered references, consensus representatives): fused default (cluster-major member
    distances, phase 4 for the listed queries, the sweep inside whole subtrees of one cluster on the static schedule of the clade
    blocks) / a thread per (query, member) pair / the listed queries through full rows / full rows + general selection for every
    query / no clade blocks (every observed leaf through the per-query merged sweep) / -c HYBRID through the level loop's per-edge
    records instead of the lean sweep's ranking (a no-op for the other criteria) / the member distances of accepted clusters by bit
    counts instead of the matrix cores.  Seed 1 is the stream that exposed the
    round-3 `k_select` fault."""
    rng = np.random.default_rng(seed)
    routes = (('default', ()), ('by_query', ('cluster_by_query',)), ('no_topup', ('no_cluster_topup',)), ('no_fuse', ('no_fuse',)),
              ('no_blocks', ('no_blocks',)), ('hybrid_records', ('hybrid_records',)), ('no_cluster_mfma', ('no_cluster_mfma',)),
              # ragged rows (csrc/common.h, Workspace::ragged; by default for references of 65 536 rows and more): small rows of 32
              # entries, so that most queries move to a big row; and two big rows only, so that a device batch runs out of them and
              # the block is placed once more with full rows
              ('ragged', {'APPLES_RAGGED_SMALL': 32}), ('ragged_dry', {'APPLES_RAGGED_SMALL': 32, 'APPLES_RAGGED_BIG': 2}))
    checked = 0
    for c in range(NCFG):  # (seed 1, first 30 configurations: the round-3 script run that found the fault)
        n = int(rng.choice([60, 257, 600, 1500, 5000, 12000])); L = int(rng.integers(40, 2047)); nq = int(rng.integers(1, 900))
        gap = float(rng.choice([0.0, 0.05, 0.3, 0.6])); thr = float(rng.choice([0.0, 0.02, 0.2, 0.5, 1.2])); b = int(rng.choice([3, 25, 200]))
        mb = int(rng.choice([0, 0, 32, 96, 160, 512])); m = str(rng.choice(['OLS', 'FM', 'BME', 'BE']))
        diam = float(rng.choice([0.01, 0.05, 0.24, 0.4, 0.8]))
        d, shp = _shaped(seed, c, n, lambda spine: synth.make_dataset(n, L, nq, gap_rate=gap, seed_tree=200 + c, mean_len=float(rng.choice([0.003, 0.01, 0.05])), spine=spine))
        nodes = np.array([d.tree.name_to_node[x] for x in d.ref_names], np.int32)
        ca = ReducedReference(Alignment(d.ref_names, d.ref_seqs), False, treecluster.grouped(d.tree, diam)).cluster_arrays()
        crit, neg = _crit(seed, c)
        if os.environ.get('APPLES_FUZZ_TRACE'):
            print('cfg', seed, c, n, L, nq, gap, thr, b, mb, m, crit, neg, diam, shp, file=sys.stderr, flush=True)
        if os.environ.get('APPLES_FUZZ_ONLY') and c != int(os.environ['APPLES_FUZZ_ONLY']):
            continue
        out = _place(routes, lambda dbg, knobs=None: Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method=m, criterion=crit, negative=neg,
                                                            threshold=thr, baseobs=b, max_batch=mb, debug=dbg, knobs=knobs), d.query_seqs)
        tag = shp + ' seed %d cfg %d: n %d L %d nq %d gap %g thr %g b %d batch %d %s %s%s diam %g' % (seed, c, n, L, nq, gap, thr, b, mb, m, crit,
                                                                                          ' -n' if neg else '', diam)
        for k in ('by_query', 'no_topup', 'no_fuse', 'no_blocks', 'hybrid_records', 'no_cluster_mfma', 'ragged', 'ragged_dry'):
            if k not in out:
                continue
            assert out[k].tobytes() == out['default'].tobytes(), '%s: default vs %s: %s' % (tag, k, _diff(out['default'], out[k]))
        if n <= 1500:
            co = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, method=m, criterion=crit, negative=neg, threshold=thr, baseobs=b,
                         lut=jc69_lut(L, 0.001), threads=NTHREADS)
            want = co.place_sequences(d.query_seqs)
            assert want.tobytes() == out['default'].tobytes(), '%s: default vs C oracle: %s' % (tag, _diff(out['default'], want))
            checked += 1
    assert checked >= 5


@pytest.mark.parametrize('seed', [3, 10] + EXTRA_SEEDS)
def test_singleton_jc69_routes_agree(seed):
    """Singleton clusters, JC69: GEMM-form fused pass + lean / bit sweep (default) against the bit-plane-fed matrix-core kernel
    with merged level lists, and against full rows + general selection with the node map: no distance, selection or
    sweep code in common; the default (top-up chain beside the sweep, from 2 048 nodes) against the chain before the sweep;
    small backbones also against the C oracle."""
    rng = np.random.default_rng(seed)
    routes = (('default', ()), ('no_gemm', ('no_dist_gemm', 'sweep_merge')), ('no_fuse', ('no_fuse', 'node_map')),
              ('no_topup', ('no_topup_kernel', 'no_sweep_lean')), ('serial_topup', ('no_topup_overlap',)))
    checked = 0
    for c in range(NCFG):
        n = int(rng.choice([40, 257, 600, 1500, 5000, 20000])); L = int(rng.integers(20, 2047)); nq = int(rng.integers(1, 700))
        gap = float(rng.choice([0.0, 0.05, 0.3, 0.6])); thr = float(rng.choice([0.05, 0.2, 0.5, 1.2])); b = int(rng.choice([3, 25, 200]))
        mb = int(rng.choice([0, 0, 32, 96, 160, 512])); m = str(rng.choice(['OLS', 'FM', 'BME', 'BE']))
        d, shp = _shaped(seed, c, n, lambda spine: synth.make_dataset(n, L, nq, gap_rate=gap, seed_tree=100 + c, mean_len=float(rng.choice([0.003, 0.01, 0.05])), spine=spine))
        nodes = np.array([d.tree.name_to_node[x] for x in d.ref_names], np.int32)
        # bytes beyond ACGT- (ordinary symbols to apples/distance.py:733-737), drawn apart from the configuration stream: none / a `.`
        # in a few reference rows and a `?` in one query / such bytes in most rows and many queries, runs of them at the rows' ends
        rx = np.random.default_rng([seed, c, 79])
        ex = int(rx.integers(0, 4))
        if ex >= 2:
            d.ref_seqs = d.ref_seqs.copy(); d.query_seqs = d.query_seqs.copy()
            rows = np.nonzero(rx.random(n) < (0.05 if ex == 2 else 0.8))[0]
            d.ref_seqs[rows, rx.integers(0, L, size=len(rows))] = ord('.')
            d.query_seqs[int(rx.integers(0, nq)), int(rx.integers(0, L))] = ord('?')
            if ex == 3:
                for i in rows[::3]:
                    d.ref_seqs[i, :int(rx.integers(0, L // 4 + 1))] = ord('.')
                for i in range(0, nq, 3):
                    d.query_seqs[i, L - int(rx.integers(0, L // 4 + 1)):] = ord(str(rx.choice(['.', '*'])))
            shp += ' exotic %d' % ex
        crit, neg = _crit(seed, c)
        out = _place(routes, lambda dbg: Engine(d.tree, d.ref_seqs, nodes, method=m, criterion=crit, negative=neg, threshold=thr,
                                                baseobs=b, max_batch=mb, debug=dbg), d.query_seqs)
        tag = shp + ' seed %d cfg %d: n %d L %d nq %d gap %g thr %g b %d batch %d %s %s%s' % (seed, c, n, L, nq, gap, thr, b, mb, m, crit, ' -n' if neg else '')
        for k in ('no_gemm', 'no_fuse', 'no_topup', 'serial_topup'):
            assert out[k].tobytes() == out['default'].tobytes(), '%s: default vs %s: %s' % (tag, k, _diff(out['default'], out[k]))
        if n <= 1500:
            co = COracle(d.tree, d.ref_seqs, nodes, method=m, criterion=crit, negative=neg, threshold=thr, baseobs=b, lut=jc69_lut(L, 0.001),
                         threads=NTHREADS)
            want = co.place_sequences(d.query_seqs)
            assert want.tobytes() == out['default'].tobytes(), '%s: default vs C oracle: %s' % (tag, _diff(out['default'], want))
            checked += 1
    assert checked >= 5


def _against_c_oracle_scoredist(got, want, tag, crit, neg):
    """scoredist placements against the C oracle's: byte for byte.  Until round 6 the distances carried the device's log where
    the oracle's carry libm's -- one unit in the last place apart now and then, which decided between mathematically tied
    candidates -- and this check allowed a bounded class of verified ties; since csrc/libm_log.h restates libm's log bit for bit
    (as pow2_libm closed the residuals) there is no tolerance left: the sums run over the sites left to right on both sides
    (the reference's own order is BLAS-internal).  Returns the number of rows that differ: 0."""
    assert got.tobytes() == want.tobytes(), '%s: default vs C oracle: %s' % (tag, _diff(got, want))
    return 0


@pytest.mark.parametrize('seed', [4, 11] + EXTRA_SEEDS)
def test_scoredist_routes_agree(seed):
    """scoredist, singleton clusters: matrix-core lower-bound filter (fp4 table values) + exact candidates + lower-bound top-up
    (default) against the same with fp6 table values, against every pair with the early exit (no filter), against the filter with full rows for the top-up list, against full
    rows + general selection, against the top-up chain before the sweep, and against the top-up's row form for every listed query / for
    nearly every one (compact lists of 16 entries: the hand-over to the row form); small backbones also against the C oracle."""
    rng = np.random.default_rng(seed)
    routes = (('default', ()), ('fp6', ('sd_fp6',)), ('every_pair', ('no_sd_gemm',)), ('rows_topup', ('no_sd_topup',)), ('no_fuse', ('no_fuse',)),
              ('serial_topup', ('no_topup_overlap',)), ('row_lists', ('no_sd_compact',)), ('tiny_lists', ('sd_compact_tiny',)))
    checked = ties = 0
    for c in range(NCFG):
        n = int(rng.choice([40, 257, 600, 1500, 5000, 20000])); L = int(rng.integers(7, 1200)); nq = int(rng.integers(1, 700))
        gap = float(rng.choice([0.0, 0.05, 0.3, 0.6])); thr = float(rng.choice([0.03, 0.1, 0.2, 0.24, 0.5])); b = int(rng.choice([3, 25, 200]))
        mb = int(rng.choice([0, 0, 32, 96, 160, 512])); m = str(rng.choice(['OLS', 'FM', 'BME', 'BE']))
        d, shp = _shaped(seed, c, n, lambda spine: synth.make_dataset(n, L, nq, protein=True, gap_rate=gap, seed_tree=400 + c, mean_len=float(rng.choice([0.003, 0.01, 0.05])), spine=spine))
        q = d.query_seqs.copy()
        if nq > 6:
            q[3] = d.ref_seqs[11 % n]          # an exact match
            q[4] = ord('-')                    # nothing observed
            q[5, ::2] = ord('x')               # symbols outside the alphabet count as 'A' (apples/distance.py:418-678)
        nodes = np.array([d.tree.name_to_node[x] for x in d.ref_names], np.int32)
        crit, neg = _crit(seed, c)
        out = _place(routes, lambda dbg: Engine(d.tree, d.ref_seqs, nodes, protein=True, method=m, criterion=crit, negative=neg,
                                                threshold=thr, baseobs=b, max_batch=mb, debug=dbg), q)
        tag = shp + ' seed %d cfg %d: n %d L %d nq %d gap %g thr %g b %d batch %d %s %s%s' % (seed, c, n, L, nq, gap, thr, b, mb, m, crit, ' -n' if neg else '')
        for k in ('fp6', 'every_pair', 'rows_topup', 'no_fuse', 'serial_topup', 'row_lists', 'tiny_lists'):
            assert out[k].tobytes() == out['default'].tobytes(), '%s: default vs %s: %s' % (tag, k, _diff(out['default'], out[k]))
        if n <= 1500:
            co = COracle(d.tree, d.ref_seqs, nodes, protein=True, method=m, criterion=crit, negative=neg, threshold=thr, baseobs=b,
                         threads=NTHREADS)
            ties += _against_c_oracle_scoredist(out['default'], co.place_sequences(q), tag, crit, neg)
            checked += 1
    print('scoredist family, seed %d: %d configurations against the C oracle, %d tie-class rows' % (seed, checked, ties))
    assert checked >= 5


@pytest.mark.parametrize('seed', [6, 12] + EXTRA_SEEDS)
def test_clustered_scoredist_routes_agree(seed):
    """The command line's default PROTEIN route (-p with clusters: scoredist to consensus representatives of the 21-symbol
    alphabet, members of the accepted clusters, the top-up rule; apples/Reference.py:117-157): the fused default (distances to the
    representatives alone, cluster-major member distances of the accepted clusters) against the same with the listed queries
    through full rows, with the queries beyond 512 accepted clusters through full rows, and against full rows + general selection
    for every query (round 3's route, the only one before round 5); small backbones also against the C oracle."""
    rng = np.random.default_rng(seed)
    routes = (('default', ()), ('no_topup', ('no_cluster_topup',)), ('no_big', ('no_cluster_big',)), ('no_fuse', ('no_fuse',)),
              ('no_blocks', ('no_blocks',)), ('ragged', {'APPLES_RAGGED_SMALL': 32}), ('ragged_dry', {'APPLES_RAGGED_SMALL': 32, 'APPLES_RAGGED_BIG': 2}))
    checked = ties = 0
    for c in range(NCFG):
        n = int(rng.choice([60, 257, 600, 1500, 5000, 12000])); L = int(rng.integers(7, 1200)); nq = int(rng.integers(1, 700))
        gap = float(rng.choice([0.0, 0.05, 0.3, 0.6])); thr = float(rng.choice([0.0, 0.03, 0.1, 0.2, 0.24, 0.5])); b = int(rng.choice([3, 25, 200]))
        mb = int(rng.choice([0, 0, 32, 96, 160, 512])); m = str(rng.choice(['OLS', 'FM', 'BME', 'BE']))
        diam = float(rng.choice([0.01, 0.05, 0.24, 0.4, 0.8]))
        d, shp = _shaped(seed, c, n, lambda spine: synth.make_dataset(n, L, nq, protein=True, gap_rate=gap, seed_tree=500 + c, mean_len=float(rng.choice([0.003, 0.01, 0.05])), spine=spine))
        q = d.query_seqs.copy()
        if nq > 6:
            q[3] = d.ref_seqs[11 % n]          # an exact match (-0.0 from scoredist)
            q[4] = ord('-')                    # nothing observed
            q[5, ::2] = ord('x')               # symbols outside the alphabet count as 'A'
        nodes = np.array([d.tree.name_to_node[x] for x in d.ref_names], np.int32)
        ca = ReducedReference(Alignment(d.ref_names, d.ref_seqs), True, treecluster.grouped(d.tree, diam)).cluster_arrays()
        crit, neg = _crit(seed, c)
        if os.environ.get('APPLES_FUZZ_TRACE'):
            print('cfg', seed, c, n, L, nq, gap, thr, b, mb, m, crit, neg, diam, shp, file=sys.stderr, flush=True)
        if os.environ.get('APPLES_FUZZ_ONLY') and c != int(os.environ['APPLES_FUZZ_ONLY']):
            continue
        out = _place(routes, lambda dbg, knobs=None: Engine(d.tree, d.ref_seqs, nodes, clusters=ca, protein=True, method=m, criterion=crit,
                                                            negative=neg, threshold=thr, baseobs=b, max_batch=mb, debug=dbg, knobs=knobs), q)
        tag = shp + ' seed %d cfg %d: n %d L %d nq %d gap %g thr %g b %d batch %d %s %s%s diam %g' % (seed, c, n, L, nq, gap, thr, b, mb, m, crit,
                                                                                          ' -n' if neg else '', diam)
        for k in ('no_topup', 'no_big', 'no_fuse', 'no_blocks', 'ragged', 'ragged_dry'):
            if k not in out:
                continue
            assert out[k].tobytes() == out['default'].tobytes(), '%s: default vs %s: %s' % (tag, k, _diff(out['default'], out[k]))
        if n <= 1500:
            co = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, protein=True, method=m, criterion=crit, negative=neg, threshold=thr,
                         baseobs=b, threads=NTHREADS)
            ties += _against_c_oracle_scoredist(out['default'], co.place_sequences(q), tag, crit, neg)
            checked += 1
    print('clustered scoredist family, seed %d: %d configurations against the C oracle, %d tie-class rows' % (seed, checked, ties))
    assert checked >= 5


@pytest.mark.parametrize('seed', [1, 5] + EXTRA_SEEDS)
def test_distance_table_routes_against_c_oracle(seed):
    """-d input (apples/PoolQueryWorker.py:44-59): odd and even numbers of columns (rows then start 8- or 16-byte aligned, which
    picks the loads of the selection kernels), columns in random order, columns that are not tree leaves, negative and zero
    entries, ties; streaming selection (default), general selection and the level-loop sweep, all byte for byte against the
    C oracle."""
    rng = np.random.default_rng(seed)
    routes = (('default', ()), ('no_stream', ('no_stream_select',)), ('no_lean', ('no_sweep_lean', 'no_topup_kernel')),
              ('third_pass', ('stream_third_pass',)))
    for c in range(NCFG):
        n = int(rng.choice([33, 64, 257, 1000, 4097, 20001])); nq = int(rng.integers(1, 200))
        thr = float(rng.choice([0.0, 0.05, 0.2, 1.0])); b = int(rng.choice([3, 25, 200])); m = str(rng.choice(['OLS', 'FM', 'BME', 'BE']))
        d, shp = _shaped(seed, c, n, lambda spine: synth.make_dataset(n, 8, nq, seed_tree=300 + c, mean_len=float(rng.choice([0.003, 0.01, 0.05])), spine=spine))
        nodes = np.array([d.tree.name_to_node[x] for x in d.ref_names], np.int32)
        D = synth.noisy_distance_rows(d.tree, d.query_leaf, d.query_pendant, list(range(nq)), seed_noise=c)
        perm = rng.permutation(n); D = np.ascontiguousarray(D[:, perm]); cols = nodes[perm].copy()
        off = rng.random(n) < float(rng.choice([0.0, 0.02, 0.3])); cols[off] = -1               # columns that are not tree leaves
        neg = rng.random(D.shape) < float(rng.choice([0.0, 0.01])); D[neg] = -1.0               # invalid entries
        zer = rng.random(D.shape) < float(rng.choice([0.0, 0.0005])); D[zer] = 0.0              # exact matches (also outside the tree)
        tie = rng.random(D.shape) < 0.01; D[tie] = np.round(D[tie], 2)                          # ties
        crit, neg = _crit(seed, c)
        want = COracle(d.tree, method=m, criterion=crit, negative=neg, threshold=thr, baseobs=b, threads=NTHREADS).place_distances(D, cols)
        out = _place(routes, lambda dbg: Engine(d.tree, None, method=m, criterion=crit, negative=neg, threshold=thr, baseobs=b, debug=dbg),
                     (D, cols), place='place_distances')
        tag = shp + ' seed %d cfg %d: n %d nq %d thr %g b %d %s %s%s off-tree %d' % (seed, c, n, nq, thr, b, m, crit, ' -n' if neg else '', int(off.sum()))
        for k in out:
            assert out[k].tobytes() == want.tobytes(), '%s: %s vs C oracle: %s' % (tag, k, _diff(out[k], want))
