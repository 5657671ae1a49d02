"""Parity of the HIP path (through the C ABI) against the golden fixtures the reference produced
and against the CPU oracle on the same seeded inputs.  Needs an MI355X: run with -m gpu."""
import os
import sys

import numpy as np
import pytest

from helpers import DATA, GOLD, ROOT, assert_prow, load_json, prow_is_tie, read_dismat

sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import apples_oracle as orc  # noqa: E402

from apples_amd import synth  # noqa: E402
from apples_amd.engine import Engine, placement_row, F_EXACT, F_INSUFFICIENT  # noqa: E402
from apples_amd.fasta import read_alignment  # noqa: E402
from apples_amd.tree import read_tree  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def c1():
    tree = read_tree(os.path.join(DATA, 'backbone.nwk'))
    ref = read_alignment(os.path.join(DATA, 'ref.fa'), False, False)
    qry = read_alignment(os.path.join(DATA, 'query.fa'), False, False)
    nodes = np.array([tree.name_to_node.get(n, -1) for n in ref.names], np.int32)
    return tree, ref, qry, nodes


@pytest.fixture(scope='module')
def eng_c1(c1):
    tree, ref, qry, nodes = c1
    e = Engine(tree, ref.seqs, nodes, method='OLS')
    yield e
    e.close()


def test_jc69_counts_and_distances(c1, eng_c1):
    tree, ref, qry, nodes = c1
    counts, dist = eng_c1.distances(qry.seqs)
    g = np.load(os.path.join(GOLD, 'g1_jc69_data.npz'))
    want = np.array([[orc.pair_counts(q, r) for r in ref.seqs] for q in qry.seqs], dtype=np.uint32)
    assert np.array_equal(counts, want)  # integers: bit exact
    np.testing.assert_allclose(dist, g['dist'], rtol=1e-6, atol=0)  # north_star tolerance
    # with the numpy-filled table the device hands out the host's own log bits
    host = np.array([[orc.jc69(q, r, 0.001) for r in ref.seqs] for q in qry.seqs])
    assert np.array_equal(dist, host)


def test_jc69_edge_cases_and_byte_symbols():
    g = np.load(os.path.join(GOLD, 'g1_jc69_synth.npz'))
    a, b = g['a'], g['b']
    tree = read_tree(os.path.join(DATA, 'small_backbone.nwk'))
    for V, key in ((0.001, 'jc69_V0.001'), (0.5, 'jc69_V0.5')):
        # rows of `b` act as the reference; '*' and '?' are ordinary symbols (apples/distance.py:733): the context keeps its
        # 2-plane rows (such bytes as gaps there) and an 8-plane copy, which full rows come from
        e = Engine(tree, b, np.full(len(b), -1, np.int32), method='OLS', overlap=V)
        info = e.describe()
        assert info['code_planes'] == 2 and info['eight_plane_copy'] == 1 and info['exotic_sites_max_per_row'] > 0, info
        counts, dist = e.distances(a)
        for i in range(len(a)):
            assert tuple(counts[i, i]) == orc.pair_counts(a[i], b[i])
            w = g[key][i]
            assert (dist[i, i] == w) or abs(dist[i, i] - w) <= 1e-6 * abs(w)
        e.close()
    # ACGT- only -> 2-plane fast path; a later query block with '*' brings the 8-plane copy into being, the context stays as it is
    keep = [i for i in range(len(b)) if not (set(b[i].tobytes()) - set(b'ACGT-'))]
    e = Engine(tree, b[keep], np.full(len(keep), -1, np.int32), method='OLS')
    assert e.describe()['code_planes'] == 2
    c1_, d1 = e.distances(a[:8])      # rows 0..7 of `a` are plain
    assert e.describe()['eight_plane_copy'] == 0
    c2_, d2 = e.distances(a[8:10])    # row 8 carries '*'
    assert e.describe()['code_planes'] == 2 and e.describe()['eight_plane_copy'] == 1
    c3_, d3 = e.distances(a[:8])
    assert np.array_equal(c1_, c3_) and np.array_equal(d1, d3)
    for qi in range(8, 10):
        for k, ri in enumerate(keep):
            assert tuple(c2_[qi - 8, k]) == orc.pair_counts(a[qi], b[ri])
    e.close()


def test_jc69_device_log_without_table(c1):
    tree, ref, qry, nodes = c1
    e = Engine(tree, ref.seqs, nodes, method='OLS', use_lut=False)
    _, dist = e.distances(qry.seqs, want_counts=False)
    g = np.load(os.path.join(GOLD, 'g1_jc69_data.npz'))
    np.testing.assert_allclose(dist, g['dist'], rtol=1e-12, atol=0)  # (the fixture carries numpy's log: its SIMD form on some CPUs)
    e.close()
    # the device's log is libm's bit for bit (csrc/libm_log.h): the C oracle, which calls libm, gives the same bytes
    from oracle_c import COracle
    want = COracle(tree, ref.seqs, nodes).distances(qry.seqs)
    assert dist.tobytes() == want.tobytes()


def test_device_log_is_libm_log_bit_for_bit():
    """csrc/libm_log.h against the host's libm on a million arguments per family: uniform in (0, 1), around 1 on both sides of
    the routine's branch cut at 1 - 2^-4, random mantissas with exponents 2^-59 .. 1, JC69-shaped 1 - 4 m / (3 v) and scoredist-shaped
    1 - tot / valid (apples/distance.py:715,745).  The host side is ctypes on libm.so.6: the routine the C oracle links."""
    import ctypes
    from apples_amd.engine import device_log
    libm = ctypes.CDLL('libm.so.6')
    libm.log.restype = ctypes.c_double
    libm.log.argtypes = [ctypes.c_double]
    rng = np.random.default_rng(5)
    n = 1 << 20
    v = rng.integers(1, 4097, size=n).astype(np.float64)
    m = np.floor(rng.random(n) * (v + 1))
    fam = [rng.random(n), 1.0 - rng.random(n) * 0.13, 1.0 + rng.random(n) * 0.07,
           np.ldexp(1.0 + rng.random(n), -rng.integers(0, 60, size=n)), 1 - (4 * (m / v) / 3), 1 - (rng.random(n) * 1.8 * v) / v,
           np.array([1.0, 0.9375, np.nextafter(0.9375, 0), np.nextafter(1.0, 0), np.nextafter(1.0, 2), float.fromhex('0x1.109p+0'),
                     np.nextafter(float.fromhex('0x1.109p+0'), 0), 0.5, 2.0 ** -52, 2.0 ** -1000])]
    x = np.concatenate(fam)
    x = np.ascontiguousarray(x[x > 0])
    got = device_log(x)
    want = np.array([libm.log(float(t)) for t in x[::37]])  # (ctypes call per value: a strided sample of the million ...)
    assert got[::37].tobytes() == want.tobytes()
    from oracle_c import libm_log_array                      # ... and every one of them through the C oracle's loop over libm's log
    assert got.tobytes() == libm_log_array(x).tobytes()


def test_scoredist_against_golden():
    g = np.load(os.path.join(GOLD, 'g1_scoredist_synth.npz'))
    a, b = g['a'], g['b']
    tree = read_tree(os.path.join(DATA, 'small_backbone.nwk'))
    for V, key in ((0.001, 'scoredist_V0.001'), (0.5, 'scoredist_V0.5')):
        e = Engine(tree, b, np.full(len(b), -1, np.int32), protein=True, overlap=V)
        counts, dist = e.distances(a)
        got = np.array([dist[i, i] for i in range(len(a))])
        want = g[key]
        assert np.array_equal(got < 0, want < 0)
        assert np.array_equal(np.signbit(got), np.signbit(want))  # identical pair -> -0.0
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-15)
        seq = np.array([orc.scoredist_sequential(a[i], b[i], V) for i in range(len(a))])
        np.testing.assert_allclose(got, seq, rtol=1e-14, atol=1e-15)  # same summation order as the kernel
        for i in range(len(a)):
            assert counts[i, i, 1] == orc.pair_counts(a[i], b[i])[1]
        e.close()


METHODS = ('OLS', 'FM', 'BME', 'BE')


def test_sweep_per_edge_against_golden(c1):
    """S/R tuples, 2x2 solutions and residuals of every candidate edge (G3)."""
    tree, ref, qry, nodes = c1
    g = np.load(os.path.join(GOLD, 'g3_per_edge.npz'))
    e = Engine(tree, None, method='OLS')
    for qi in range(3):
        names = [str(x) for x in g['q%d_obs_names' % qi]]
        on = np.array([tree.name_to_node[k] for k in names], np.int32)
        od = g['q%d_obs_dist' % qi]
        for m in METHODS:
            e.set_options(method=m)
            r = e.sweep_edges(on, od)
            key = 'q%d_%s_' % (qi, m)
            edges = g[key + 'edge']
            assert np.array_equal(np.nonzero(r['valid'])[0], edges)
            assert r['lca'] == int(g[key + 'lca']) and r['placement']['n_valid'] == int(g[key + 'num_nodes'])
            # same IEEE operations in the same order: bit identical
            assert np.array_equal(r['S'][edges], g[key + 'S']), (qi, m)
            assert np.array_equal(r['R'][edges], g[key + 'R']), (qi, m)
            assert np.array_equal(r['x'][edges], g[key + 'x']), (qi, m)
            # residual: `x ** 2` is libm pow in the reference; the kernel's pow2_libm reproduces its bits
            assert np.array_equal(r['err'][edges], g[key + 'err']), (qi, m)
    e.close()


def _rows(eng, seqs, self_rows=None):
    return [placement_row(p) for p in eng.place_sequences(seqs, self_rows)]


def test_placements_alignment_golden(c1):
    tree, ref, qry, nodes = c1
    g = load_json('g4_placements.json')
    eng = Engine(tree, ref.seqs, nodes)
    for case in g['aln']:
        if case.get('clusters') == 'clades':
            continue
        eng.set_options(method=case['m'], criterion=case['c'], negative=case['n'], threshold=case['f'],
                        baseobs=case['b'])
        got = _rows(eng, qry.seqs)
        for row, w in zip(got, case['p']):
            assert_prow(row, w['p'], ctx='aln %s/%s/n=%s/f=%s %s' % (case['m'], case['c'], case['n'], case['f'], w['n']))
    eng.close()


def test_placements_clustered_reference_golden(c1):
    tree, ref, qry, nodes = c1
    reps = load_json('g2_selection.json')['clade_clusters']
    cons, rep_row, moff, mrow = [], [], [0], []
    for r in reps:
        if len(r['members']) == 1 and ref.seqs[ref.index[r['members'][0]]].tobytes().decode() == r['cons']:
            rep_row.append(ref.index[r['members'][0]])
        else:
            rep_row.append(len(ref) + len(cons))
            cons.append(np.frombuffer(r['cons'].encode(), np.uint8))
        mrow += [ref.index[m] for m in r['members']]
        moff.append(len(mrow))
    clusters = (np.array(cons, np.uint8).reshape(-1, ref.length), rep_row, moff, mrow)
    eng = Engine(tree, ref.seqs, nodes, clusters=clusters)
    assert eng.describe()['all_singleton'] == 0
    g = load_json('g4_placements.json')
    n = 0
    for case in g['aln']:
        if case.get('clusters') != 'clades':
            continue
        eng.set_options(method=case['m'], criterion=case['c'], negative=case['n'], threshold=case['f'], baseobs=case['b'])
        for row, w in zip(_rows(eng, qry.seqs), case['p']):
            assert_prow(row, w['p'], ctx='clades %s %s' % (case['m'], w['n']))
        n += 1
    assert n == 2
    # the observed set itself, against get_obs_dist's dict (G2): compare through n_obs
    sel = load_json('g2_selection.json')
    for case in sel['cases']:
        if case['clusters'] != 'clades':
            continue
        eng.set_options(method='OLS', threshold=case['f'], baseobs=case['b'])
        p = eng.place_sequences(qry.seqs[qry.index[case['query']]][None, :])[0]
        assert p['n_obs'] == len(case['obs']), case
    eng.close()


def _cluster_arrays(reps, ref):
    """(consensus rows, rep_row, member_off, member_row) of a fixture's representative list [{cons, members}]"""
    cons, rep_row, moff, mrow = [], [], [0], []
    for r in reps:
        if len(r['members']) == 1 and ref.seqs[ref.index[r['members'][0]]].tobytes().decode() == r['cons']:
            rep_row.append(ref.index[r['members'][0]])
        else:
            rep_row.append(len(ref) + len(cons))
            cons.append(np.frombuffer(r['cons'].encode(), np.uint8))
        mrow += [ref.index[m] for m in r['members']]
        moff.append(len(mrow))
    return np.array(cons, np.uint8).reshape(-1, ref.length), rep_row, moff, mrow


@pytest.mark.parametrize('route', ['default', 'no_fuse'])
def test_protein_clustered_reference_golden(route, tmp_path):
    """The command line's default protein route, -p with clusters: scoredist to consensus representatives of the 21-symbol
    alphabet, members of the accepted clusters, the top-up rule (apples/Reference.py:117-157, apples/PoolRepresentativeWorker.py:33-58)
    against what the reference itself returned for the same clusters (g10): placements for 4 methods, 3 criteria, -n, two
    (f, b) pairs, and the size of every observed dict."""
    import prot_cases
    from apples_amd.reference import consensus
    ref_fp, qry_fp, tree_fp = prot_cases.write_case(str(tmp_path))
    tree = read_tree(tree_fp)
    ref = read_alignment(ref_fp, True, False)
    qry = read_alignment(qry_fp, True, False)
    nodes = np.array([tree.name_to_node.get(n, -1) for n in ref.names], np.int32)
    g = load_json('g10_prot_clustered.json')
    # the consensus rows this build forms are the reference's
    for r in g['clade_clusters']:
        if len(r['members']) > 1:
            assert consensus(ref.seqs[[ref.index[m] for m in r['members']]], True).tobytes().decode() == r['cons']
    eng = Engine(tree, ref.seqs, nodes, clusters=_cluster_arrays(g['clade_clusters'], ref), protein=True,
                 debug=() if route == 'default' else (route,))
    info = eng.describe()
    assert info['all_singleton'] == 0
    if route == 'default':
        assert info['cluster_fused'] == 1, info
    ties = 0
    for case in g['placements']:
        eng.set_options(method=case['m'], criterion=case['c'], negative=case['n'], threshold=case['f'], baseobs=case['b'])
        for row, w in zip(_rows(eng, qry.seqs), case['p']):
            if prow_is_tie(row, w['p']):  # (another of the edges that meet at the attachment point: same residual, same pendant)
                ties += 1
                continue
            assert_prow(row, w['p'], ctx='prot clades %s/%s/n=%s/f=%s %s' % (case['m'], case['c'], case['n'], case['f'], w['n']))
    print('protein clustered golden (%s): %d tie-class rows of %d' % (route, ties, 24 * len(g['placements'])))
    assert ties <= 4
    n = 0
    for case in g['cases']:
        if case['clusters'] != 'clades':
            continue
        eng.set_options(method='FM', criterion='MLSE', negative=False, threshold=case['f'], baseobs=case['b'])
        p = eng.place_sequences(qry.seqs[qry.index[case['query']]][None, :])[0]
        assert p['n_obs'] == len(case['obs']), (case['query'], case['f'], case['b'])
        n += 1
    assert n == 24
    eng.close()
    # singleton clusters on the same inputs (the dict sizes of g10's singleton cases)
    eng = Engine(tree, ref.seqs, nodes, protein=True, debug=() if route == 'default' else (route,))
    for case in g['cases']:
        if case['clusters'] == 'singleton':
            eng.set_options(threshold=case['f'], baseobs=case['b'])
            p = eng.place_sequences(qry.seqs[qry.index[case['query']]][None, :])[0]
            assert p['n_obs'] == len(case['obs']), (case['query'], case['f'], case['b'])
    eng.close()


def test_placements_distance_table_golden(c1):
    tree, ref, qry, nodes = c1
    g = load_json('g4_placements.json')
    rows = list(read_dismat(os.path.join(DATA, 'dist.mat')))
    cols = list(rows[0][1])
    D = np.array([[r[1][c] for c in cols] for r in rows])
    col_nodes = np.array([tree.name_to_node.get(c, -1) for c in cols], np.int32)
    eng = Engine(tree, None)
    for case in g['dist']:
        eng.set_options(method=case['m'], threshold=case['f'], baseobs=case['b'])
        got = [placement_row(p) for p in eng.place_distances(D, col_nodes)]
        for row, w in zip(got, case['p']):
            assert_prow(row, w['p'], ctx='-d %s f=%s %s' % (case['m'], case['f'], w['n']))
    eng.close()
    stree = read_tree(os.path.join(DATA, 'small_backbone.nwk'))
    srows = list(read_dismat(os.path.join(DATA, 'small_dist.mat')))
    scols = list(srows[0][1])
    sD = np.array([[r[1][c] for c in scols] for r in srows])
    eng = Engine(stree, None)
    for case in g['small']:
        eng.set_options(method=case['m'])
        got = [placement_row(p) for p in eng.place_distances(sD, [stree.name_to_node[c] for c in scols])]
        assert got[0][0] == 3
        for row, w in zip(got, case['p']):
            assert_prow(row, w['p'], ctx='small %s' % case['m'])
    eng.close()


def test_edge_cases_golden(c1):
    tree, ref, qry, nodes = c1
    g = load_json('g4_placements.json')['edge_cases']
    L = ref.length
    allgap = np.full(L, ord('-'), np.uint8)
    names = ['allgap', ref.names[0], 'copy_of_second', 'allgap2', 'normal']
    assert names == g['names']
    seqs = np.vstack([allgap, ref.seqs[0], ref.seqs[1], allgap, qry.seqs[0]])
    # runquery drops the query's own entry when its name is a backbone leaf (PoolQueryWorker.py:63-66)
    self_rows = np.array([ref.index[n] if (n in ref.index and n in tree.name_to_node) else -1 for n in names], np.int32)
    eng = Engine(tree, ref.seqs, nodes, method='OLS')
    out = eng.place_sequences(seqs, self_rows)
    assert out[0]['flags'] & F_INSUFFICIENT and out[2]['flags'] & F_EXACT and out[3]['flags'] & F_INSUFFICIENT
    for p, w in zip(out, g['results']):
        assert_prow(placement_row(p), w['p'], ctx='edge case %s' % w['n'])
    eng.close()


@pytest.mark.parametrize('label', ['nt_OLS', 'aa_FM'])
def test_synthetic_alignment_golden(label):
    g = load_json('g6_synthetic.json')[label]
    d = synth.make_dataset(g['N'], g['L'], g['Q'], protein=g['protein'])
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    eng = Engine(d.tree, d.ref_seqs, nodes, protein=g['protein'], method=g['m'], threshold=g['f'], baseobs=g['b'])
    out = eng.place_sequences(d.query_seqs)
    ties = 0
    for i, p in enumerate(out):
        assert p['n_obs'] == g['n_obs'][i]
        w, bs = g['p'][i], g['best_second'][i]
        row = placement_row(p)
        if row[0] != w['p'][0] and bs is not None and abs(bs[1] - bs[0]) <= 1e-12 * max(abs(bs[0]), 1e-300):
            ties += 1  # documented tie class
            continue
        assert_prow(row, w['p'], ctx='%s q%d' % (label, i))
    assert ties <= 2
    eng.close()


@pytest.mark.parametrize('m', ['BME', 'OLS'])
def test_synthetic_distance_table_golden(m):
    g = load_json('g6_synthetic.json')['dmat_' + m]
    d = synth.make_dataset(g['N'], 500, g['Q'])
    D = synth.noisy_distance_rows(d.tree, d.query_leaf, d.query_pendant, list(range(g['Q'])))
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    eng = Engine(d.tree, None, method=m, threshold=g['f'], baseobs=g['b'])
    out = eng.place_distances(D, nodes)
    for i, p in enumerate(out):
        assert_prow(placement_row(p), g['p'][i]['p'], ctx='dmat %s q%d' % (m, i))
    eng.close()


def test_polytomies_negative_and_zero_edges_against_c_oracle():
    """data/prot/backbone.nwk: polytomies up to degree 42, 378 negative and 702 zero branch
    lengths.  Exercises the CSR child paths of the sweep and util.solve2_2's branch table on
    negative edge lengths, for every method and criterion, against the C oracle."""
    from oracle_c import COracle
    tree = read_tree(os.path.join(DATA, 'prot', 'backbone.nwk'))
    assert np.diff(tree.child_off).max() >= 40 and (tree.edge_len < 0).sum() > 300
    leaves = tree.leaves
    rng = np.random.default_rng(5)
    nq = 48
    D = np.full((nq, len(leaves)), -1.0)
    for i in range(nq):
        k = [3, 4, 25, 200, 1500, len(leaves)][i % 6]
        # observed leaves: a contiguous clade-like run plus scattered ones
        start = rng.integers(0, len(leaves) - k + 1)
        sel = np.unique(np.concatenate([np.arange(start, start + (k + 1) // 2),
                                        rng.choice(len(leaves), size=k // 2, replace=False)]))
        D[i, sel] = rng.uniform(0.01, 1.5, size=len(sel))
    cols = leaves.astype(np.int32)
    for m in METHODS:
        for c in ('MLSE', 'ME', 'HYBRID'):
            for neg in (False, True):
                eng = Engine(tree, None, method=m, criterion=c, negative=neg, threshold=10.0, baseobs=5)
                got = eng.place_distances(D, cols)
                want = COracle(tree, method=m, criterion=c, negative=neg, threshold=10.0, baseobs=5).place_distances(D, cols)
                assert np.array_equal(got['n_obs'], want['n_obs']) and np.array_equal(got['n_valid'], want['n_valid'])
                # zero-length edges make exact ties between an edge and its neighbours common here: only
                # bit-identical residuals (libm pow bits included) and the first-minimum rule reproduce them
                assert got.tobytes() == want.tobytes(), (m, c, neg, np.nonzero(got['edge'] != want['edge'])[0])
                eng.close()
    # per-edge arrays on one big observed set, bit for bit (S, R, x)
    sel = np.nonzero(D[5] >= 0)[0]
    for m in METHODS:
        eng = Engine(tree, None, method=m)
        r = eng.sweep_edges(cols[sel], D[5, sel])
        w = COracle(tree, method=m).sweep_edges(cols[sel], D[5, sel])
        assert np.array_equal(r['valid'], w['valid']) and r['lca'] == w['lca']
        v = r['valid']
        assert np.array_equal(r['S'][v], w['S'][v]) and np.array_equal(r['R'][v], w['R'][v])
        assert np.array_equal(r['x'][v], w['x'][v])
        assert np.array_equal(r['err'][v], w['err'][v])  # libm pow(x, 2) bits on both sides
        eng.close()


def _random_newick(rng, n_leaves, shape):
    """Random rooted tree text: 'random' joins, 'caterpillar', or 'bushy' (polytomies), with a few
    zero, tiny and missing branch lengths."""
    def blen():
        u = rng.random()
        if u < 0.08:
            return ':0'
        if u < 0.12:
            return ':%.3g' % (rng.random() * 1e-7)
        return ':%.5f' % rng.exponential(0.05)
    nodes = ['L%d%s' % (i, blen()) for i in range(n_leaves)]
    if shape == 'caterpillar':
        cur = nodes[0]
        for nd in nodes[1:]:
            cur = '(%s,%s)%s' % (cur, nd, blen())
        return cur.rsplit(':', 1)[0] + ';'
    while len(nodes) > 1:
        k = 2 if shape == 'random' else int(rng.integers(2, 6))
        k = min(k, len(nodes))
        idx = sorted(rng.choice(len(nodes), size=k, replace=False), reverse=True)
        kids = [nodes.pop(i) for i in idx]
        nodes.append('(%s)%s' % (','.join(kids), blen()))
    return nodes[0].rsplit(':', 1)[0] + ';'


@pytest.mark.parametrize('layout', ['bits', 'map', 'merge'])
def test_random_trees_per_edge_bit_parity_with_c_oracle(layout, monkeypatch):
    """Many small random trees (binary, caterpillar, polytomous; zero and tiny branch lengths) and
    random observed sets of every size from 2 up: valid set, LCA, S, R and the 2x2 solutions must
    equal the C oracle's bit for bit, placements edge for edge (ties resolved by residual).  Both
    node-lookup layouts of the sweep (LDS bit space; tagged node map of big trees) and the merged level lists that
    big trees get by default (forced here)."""
    from oracle_c import COracle
    if layout == 'map':
        monkeypatch.setenv('APPLES_NODE_MAP', '1')
    if layout == 'merge':
        monkeypatch.setenv('APPLES_SWEEP_MERGE', '1')
    from apples_amd.tree import parse_newick
    rng = np.random.default_rng(2024)
    n_cases = 0
    for shape in ('random', 'caterpillar', 'bushy'):
        for n_leaves in (3, 4, 7, 16, 61, 200):
            tree = parse_newick(_random_newick(rng, n_leaves, shape))
            leaves = tree.leaves
            eng = Engine(tree, None, method='OLS')
            if layout == 'merge':
                # (before a workspace exists a binary tree reports the lean form its plain passes would get; per-edge
                # inspection, used below, runs the level loop with merged lists)
                assert eng.describe()['sweep_layout'] in ('merge', 'lean')
            for m in METHODS:
                eng.set_options(method=m, criterion='MLSE')
                co = COracle(tree, method=m)
                for k in sorted({2, 3, min(5, n_leaves), n_leaves // 2 + 1, n_leaves}):
                    if k > n_leaves or k < 2:
                        continue
                    sel = np.sort(rng.choice(n_leaves, size=k, replace=False))
                    D = rng.uniform(0.02, 1.0, size=k)
                    r = eng.sweep_edges(leaves[sel], D)
                    w = co.sweep_edges(leaves[sel], D)
                    assert np.array_equal(r['valid'], w['valid']) and r['lca'] == w['lca'], (shape, n_leaves, m, k)
                    v = r['valid']
                    assert np.array_equal(r['S'][v], w['S'][v]), (shape, n_leaves, m, k)
                    assert np.array_equal(r['R'][v], w['R'][v]), (shape, n_leaves, m, k)
                    assert np.array_equal(r['x'][v], w['x'][v]), (shape, n_leaves, m, k)
                    assert np.array_equal(r['err'][v], w['err'][v]), (shape, n_leaves, m, k)
                    assert r['placement']['n_valid'] == w['placement']['n_valid']
                    assert r['placement']['edge'] == w['placement']['edge'], (shape, n_leaves, m, k)
                    n_cases += 1
            eng.close()
    assert n_cases > 250


def test_both_sweep_layouts_agree_on_polytomies(monkeypatch):
    """The sweep knows which nodes are in a query's subtree from merged level lists without tree records (the lean sweep: the
    default from 2 048 nodes, polytomies through child records), from a bit space in LDS (small
    trees) or from a tagged node map in global scratch (big trees).  Same polytomous tree, same
    observed sets, every method, HYBRID included (it looks nodes up again after the sweep): the two
    layouts must return the same bytes."""
    tree = read_tree(os.path.join(DATA, 'prot', 'backbone.nwk'))
    leaves = tree.leaves
    rng = np.random.default_rng(11)
    nq = 36
    D = np.full((nq, len(leaves)), -1.0)
    for i in range(nq):
        k = [3, 7, 60, 900, len(leaves)][i % 5]
        sel = rng.choice(len(leaves), size=k, replace=False)
        D[i, sel] = rng.uniform(0.01, 1.5, size=k)
    cols = leaves.astype(np.int32)
    for m, c in (('OLS', 'MLSE'), ('BME', 'HYBRID'), ('FM', 'ME'), ('BE', 'HYBRID')):
        outs = []
        for layout in ('lean', 'bits', 'map', 'scan', 'merge'):
            monkeypatch.delenv('APPLES_NODE_MAP', raising=False)
            monkeypatch.delenv('APPLES_SWEEP_SCAN', raising=False)
            monkeypatch.delenv('APPLES_SWEEP_MERGE', raising=False)
            monkeypatch.delenv('APPLES_NO_SWEEP_LEAN', raising=False)
            if layout != 'lean':  # (the default on this tree since round 6: the lean sweep's child records for its 77 polytomies)
                monkeypatch.setenv('APPLES_NO_SWEEP_LEAN', '1')
            if layout == 'scan':
                monkeypatch.setenv('APPLES_SWEEP_SCAN', '1')
            if layout == 'map':
                monkeypatch.setenv('APPLES_NODE_MAP', '1')
            if layout == 'merge':  # (big trees' default: level lists by merging; polytomies look their children up by search)
                monkeypatch.setenv('APPLES_SWEEP_MERGE', '1')
            eng = Engine(tree, None, method=m, criterion=c, threshold=10.0, baseobs=5)
            assert eng.describe()['sweep_layout'] == layout
            outs.append(eng.place_distances(D, cols))
            eng.close()
        assert outs[0].tobytes() == outs[1].tobytes() == outs[2].tobytes() == outs[3].tobytes() == outs[4].tobytes(), (m, c)
    monkeypatch.delenv('APPLES_NO_SWEEP_LEAN', raising=False)
    monkeypatch.delenv('APPLES_NODE_MAP', raising=False)
    monkeypatch.delenv('APPLES_SWEEP_SCAN', raising=False)
    monkeypatch.delenv('APPLES_SWEEP_MERGE', raising=False)


@pytest.mark.parametrize('L,n_ref,n_q', [(77, 130, 16), (1000, 300, 100), (1620, 257, 300), (33, 64, 517), (64, 129, 17),
                                          (4099, 70, 40)])
def test_matrix_core_pair_counts_equal_the_bytewise_definition(L, n_ref, n_q, monkeypatch):
    """The tiled distance pass counts (mismatches, shared valid sites) with fp4 MFMAs from
    tetrahedral codes; apples/distance.py:733-737 defines the same two integers on bytes.  Ragged
    sizes (L not a multiple of 32 or of the 64-site MFMA block, rows and queries not multiples of the 128 x 256 tile), heavy
    gaps, all-gap rows and identical rows; every count must be identical, and so must the counts of
    the bit-plane VALU kernel the library uses for small query tiles."""
    rng = np.random.default_rng(L * 1000 + n_q)
    alpha = np.frombuffer(b'ACGT-', dtype=np.uint8)
    ref = alpha[rng.choice(5, size=(n_ref, L), p=[0.22, 0.22, 0.22, 0.22, 0.12])]
    qry = alpha[rng.choice(5, size=(n_q, L), p=[0.2, 0.2, 0.2, 0.2, 0.2])]
    ref[0] = ord('-')
    qry[1] = ord('-')
    qry[2] = ref[5]
    tree = read_tree(os.path.join(DATA, 'small_backbone.nwk'))
    monkeypatch.setenv('APPLES_DIST_MFMA_ROWS', '1')  # full rows normally come from the bit-plane kernel
    monkeypatch.setenv('APPLES_NO_DIST_GEMM', '1')    # (the GEMM form keeps compact images that this kernel does not read)
    e = Engine(tree, ref, np.full(n_ref, -1, np.int32), method='OLS')
    assert e.describe()['code_planes'] == 2
    counts, dist = e.distances(qry)  # >= 16 queries: matrix-core kernel
    nd_r, nd_q = ref != ord('-'), qry != ord('-')
    valid = (nd_q[:, None, :] & nd_r[None, :, :]).sum(-1)
    mism = ((qry[:, None, :] != ref[None, :, :]) & nd_q[:, None, :] & nd_r[None, :, :]).sum(-1)
    want = np.stack([mism, valid], axis=-1).astype(np.uint32)  # Engine.distances: [..., 0] mismatches, [..., 1] valid
    assert np.array_equal(counts, want)
    few, _ = e.distances(qry[:7])  # < 16 queries: bit-plane kernel
    assert np.array_equal(few, want[:7])
    e.close()


def test_gemm_form_of_the_fused_pass_at_many_lengths():
    """dist_gemm.hip: every tail shape of its three-generation main loop ((2G - 2) % 3 = 0, 1, 2), the shortest
    alignment (two 64-site steps), the longest it takes (2046 sites; 2047 goes to the bit-plane-fed kernel), heavy
    gaps (small valid counts: the overlap rule and the linear threshold's lower end).  Placements must be byte-
    identical to the bit-plane-fed matrix-core kernel (APPLES_NO_DIST_GEMM), to the table form of the threshold
    (APPLES_GEMM_TABLE) and to the full-row route (APPLES_NO_FUSE), which share no distance code with it."""
    import subprocess
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from apples_amd import synth\n"
            "from apples_amd.engine import Engine\n"
            "out = []\n"
            "for L, gap, thr in ((64, 0.05, 0.3), (100, 0.3, 0.2), (300, 0.05, 0.2), (450, 0.5, 0.25), (700, 0.1, 0.15), (900, 0.05, 0.2),\n"
            "                    (1620, 0.2, 0.2), (2046, 0.05, 0.1), (2047, 0.05, 0.1)):\n"
            "    d = synth.make_dataset(700, L, 300, gap_rate=gap, seed_tree=L)\n"
            "    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)\n"
            "    e = Engine(d.tree, d.ref_seqs, nodes, method='FM', threshold=thr, baseobs=10)\n"
            "    out.append(e.place_sequences(d.query_seqs).tobytes()); e.close()\n"
            "    if L in (300, 900):  # device batches that start inside a 256-row image tile (rows 96, 192, 288)\n"
            "        e = Engine(d.tree, d.ref_seqs, nodes, method='FM', threshold=thr, baseobs=10, max_batch=96)\n"
            "        out.append(e.place_sequences(d.query_seqs).tobytes()); e.close()\n"
            "        assert out[-1] == out[-2]\n"
            "sys.stdout.buffer.write(b''.join(out))\n" % ROOT)
    outs = []
    for env in ({}, {'APPLES_NO_DIST_GEMM': '1'}, {'APPLES_GEMM_TABLE': '1'}, {'APPLES_NO_FUSE': '1'}, {'APPLES_GEMM_QT': '128'}):
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, env=dict(os.environ, **env), timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(r.stdout)
    assert len(outs[0]) == 11 * 300 * 40
    assert all(o == outs[0] for o in outs)


def test_node_numbering_other_than_post_order_is_served_without_merged_lists(monkeypatch):
    """The merged level lists of the sweep (the default of big trees, forced here) rely on left-to-right post-order
    node ids, which apples_amd/tree.py produces as the reference does (apples/util.py:57-69).  A C-ABI caller with
    another numbering must not get silently wrong placements: the context falls back to the node bits / node map, and
    the placements are the same edges under the relabelling."""
    import types
    monkeypatch.setenv('APPLES_SWEEP_MERGE', '1')
    d = synth.make_dataset(700, 200, 48)
    t = d.tree
    n = t.n_nodes
    rng = np.random.default_rng(5)
    pi = np.concatenate([rng.permutation(n - 1), [n - 1]]).astype(np.int32)   # new id of every node; the root stays last
    inv = np.empty(n, np.int32)
    inv[pi] = np.arange(n, dtype=np.int32)
    parent = np.full(n, -1, np.int32)
    parent[pi[:n - 1]] = pi[np.asarray(t.parent)[:n - 1]]
    edge_len = np.zeros(n)
    edge_len[pi] = np.asarray(t.edge_len)
    level = np.zeros(n, np.int32)
    level[pi] = np.asarray(t.level)
    counts = np.zeros(n, np.int64)
    counts[pi] = np.diff(np.asarray(t.child_off))
    child_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    child_idx = np.empty(n - 1, np.int32)
    for v in range(n):  # children keep their file order
        kids = np.asarray(t.child_idx)[t.child_off[v]:t.child_off[v + 1]]
        child_idx[child_off[pi[v]]:child_off[pi[v]] + len(kids)] = pi[kids]
    relabelled = types.SimpleNamespace(n_nodes=n, parent=parent, edge_len=edge_len, child_off=child_off, child_idx=child_idx,
                                       level=level)
    nodes = np.array([t.name_to_node[x] for x in d.ref_names], np.int32)
    e0 = Engine(t, d.ref_seqs, nodes, method='OLS')
    want = e0.place_sequences(d.query_seqs)
    assert e0.describe()['sweep_layout'] in ('merge', 'lean')
    e0.close()
    e1 = Engine(relabelled, d.ref_seqs, pi[nodes], method='OLS')
    got = e1.place_sequences(d.query_seqs)
    assert e1.describe()['sweep_layout'] == 'bits'
    e1.close()
    placed = want['edge'] >= 0
    assert np.array_equal(got['edge'][placed], pi[want['edge'][placed]])
    assert np.array_equal(got['edge'][~placed], want['edge'][~placed])
    for f in ('error', 'distal', 'pendant', 'n_obs', 'n_valid', 'flags'):
        assert np.array_equal(got[f], want[f]), f
