"""Pin the CPU oracle (oracle/apples_oracle.py) against the fixtures the REFERENCE produced
(tests/golden/make_goldens.py).  CPU only."""
import os
import sys

import numpy as np
import pytest

from helpers import DATA, GOLD, ROOT, assert_prow, load_json, read_dismat

sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import apples_oracle as orc  # noqa: E402

from apples_amd.fasta import read_alignment  # noqa: E402
from apples_amd.tree import read_tree, parse_newick, extended_newick  # noqa: E402
from apples_amd import synth  # noqa: E402


@pytest.fixture(scope='module')
def c1():
    tree = read_tree(os.path.join(DATA, 'backbone.nwk'))
    ref = read_alignment(os.path.join(DATA, 'ref.fa'), False, False)
    qry = read_alignment(os.path.join(DATA, 'query.fa'), False, False)
    return tree, ref, qry


def _same_dist(a, b):
    # -1.0 / 0.0 sentinels exact; logs may differ in the last bit across CPUs (numpy SIMD log)
    a, b = np.asarray(a), np.asarray(b)
    assert np.array_equal(a < 0, b < 0)
    assert np.array_equal(a == 0, b == 0)
    assert np.array_equal(np.signbit(a), np.signbit(b))
    np.testing.assert_allclose(a, b, rtol=1e-13, atol=0)


def test_g1_jc69_data(c1):
    tree, ref, qry = c1
    g = np.load(os.path.join(GOLD, 'g1_jc69_data.npz'))
    assert list(g['ref_names']) == ref.names and list(g['query_names']) == qry.names
    d = np.array([[orc.jc69(q, r, 0.001) for r in ref.seqs] for q in qry.seqs])
    _same_dist(d, g['dist'])
    # independent 8-decimal check: data/dist.mat is the JC69 matrix of query x extended_ref
    for qi, (qn, row) in enumerate(read_dismat(os.path.join(DATA, 'dist.mat'))):
        assert qn == qry.names[qi]
        for ri, rn in enumerate(ref.names):
            assert abs(row[rn] - d[qi, ri]) < 6e-9


def test_g1_jc69_synth():
    g = np.load(os.path.join(GOLD, 'g1_jc69_synth.npz'))
    for V, key in ((0.001, 'jc69_V0.001'), (0.5, 'jc69_V0.5')):
        d = [orc.jc69(a, b, V) for a, b in zip(g['a'], g['b'])]
        _same_dist(d, g[key])
    assert g['jc69_V0.001'][3] == -1.0 and g['jc69_V0.001'][5] == 0.0 and g['jc69_V0.001'][7] == -1.0


def test_g1_scoredist_synth():
    g = np.load(os.path.join(GOLD, 'g1_scoredist_synth.npz'))
    for V, key in ((0.001, 'scoredist_V0.001'), (0.5, 'scoredist_V0.5')):
        want = g[key]
        d = np.array([orc.scoredist(a, b, V) for a, b in zip(g['a'], g['b'])])
        ds = np.array([orc.scoredist_sequential(a, b, V) for a, b in zip(g['a'], g['b'])])
        assert np.array_equal(d < 0, want < 0) and np.array_equal(ds < 0, want < 0)
        np.testing.assert_allclose(d, want, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(ds, want, rtol=1e-12, atol=1e-15)
    assert g['scoredist_V0.001'][0] == 0 and np.signbit(g['scoredist_V0.001'][0])  # identical -> -0.0


def test_g1_encoding():
    g = load_json('g1_encoding.json')
    # rows have different lengths in this file, so compare record by record
    from apples_amd.fasta import read_records, encode_sequence
    for prot in (False, True):
        for mask in (False, True):
            with open(os.path.join(GOLD, 'g1_encoding.fa')) as f:
                got = {n: encode_sequence(s, prot, mask).tobytes().decode() for n, s in read_records(f)}
            assert got == g['prot%d_mask%d' % (prot, mask)]


def _reps_from_json(reps):
    return [(np.frombuffer(r['cons'].encode(), dtype=np.uint8), r['members']) for r in reps]


def test_g2_selection(c1):
    tree, ref, qry = c1
    g = load_json('g2_selection.json')
    rows = {n: ref.seqs[i] for i, n in enumerate(ref.names)}
    reps_single = [(ref.seqs[i], [n]) for i, n in enumerate(ref.names)]
    reps_clades = _reps_from_json(g['clade_clusters'])
    for case in g['cases']:
        reps = reps_single if case['clusters'] == 'singleton' else reps_clades
        q = qry.seqs[qry.index[case['query']]]
        obs = orc.get_obs_dist(q, reps, rows, orc.jc69, case['f'], case['b'], 0.001)
        assert list(obs) == [k for k, _ in case['obs']], case['query']  # insertion order too
        np.testing.assert_allclose(list(obs.values()), [v for _, v in case['obs']], rtol=1e-13)


def test_g3_per_edge(c1):
    tree, ref, qry = c1
    g = np.load(os.path.join(GOLD, 'g3_per_edge.npz'))
    for qi in range(3):
        names = [str(x) for x in g['q%d_obs_names' % qi]]
        dist = g['q%d_obs_dist' % qi]
        obs = dict(zip(names, dist.tolist()))
        nodes = [tree.name_to_node[k] for k in names]
        leaf_dist = {tree.name_to_node[k]: v for k, v in obs.items()}
        valid, lca, num = orc.induced_subtree(tree, nodes)
        for m in ('OLS', 'FM', 'BME', 'BE'):
            key = 'q%d_%s_' % (qi, m)
            assert int(g[key + 'lca']) == lca and int(g[key + 'num_nodes']) == num
            edges = g[key + 'edge']
            assert np.array_equal(np.nonzero(valid)[0], edges)  # valid post-order == ascending edge_index
            S = orc.s_values(tree, valid, leaf_dist, m)
            R = orc.r_values(tree, valid, lca, S, m)
            pe = orc.per_edge(tree, valid, S, R, m, False)
            # same inputs, same IEEE operations in the same order -> bit-identical
            assert np.array_equal(np.array([S[v] for v in edges], dtype=float), g[key + 'S'])
            assert np.array_equal(np.array([R[v] for v in edges], dtype=float), g[key + 'R'])
            assert np.array_equal(np.array([pe[v][:4] for v in edges], dtype=float), g[key + 'x'])
            assert np.array_equal(np.array([pe[v][4] for v in edges], dtype=float), g[key + 'err'])


def _run_aln(tree, ref, qnames, qseqs, m, c, neg, f, b, reps=None, exclude=False):
    rows = {n: ref.seqs[i] for i, n in enumerate(ref.names)}
    reps = reps if reps is not None else [(ref.seqs[i], [n]) for i, n in enumerate(ref.names)]
    out = []
    for n, s in zip(qnames, qseqs):
        obs = orc.get_obs_dist(s, reps, rows, orc.jc69, f, b, 0.001)
        out.append(orc.runquery(tree, n, obs, m, c, neg, exclude))
    return out


def _check(results, want, ctx):
    assert len(results) == len(want)
    for r, w in zip(results, want):
        pl = r['placements'][0]
        assert pl['n'][0] == w['n'], ctx
        assert_prow(pl['p'][0], w['p'], ctx=ctx + ' ' + w['n'])


def test_g4_alignment_placements(c1):
    tree, ref, qry = c1
    g = load_json('g4_placements.json')
    assert extended_newick(tree) == g['tree']
    reps_clades = _reps_from_json(load_json('g2_selection.json')['clade_clusters'])
    for case in g['aln']:
        reps = reps_clades if case.get('clusters') == 'clades' else None
        res = _run_aln(tree, ref, qry.names, qry.seqs, case['m'], case['c'], case['n'], case['f'], case['b'], reps)
        _check(res, case['p'], 'aln %s/%s/n=%s/f=%s' % (case['m'], case['c'], case['n'], case['f']))


def test_g4_distance_table(c1):
    tree, ref, qry = c1
    g = load_json('g4_placements.json')
    rows = list(read_dismat(os.path.join(DATA, 'dist.mat')))
    for case in g['dist']:
        res = []
        for qn, obs in rows:
            obs = orc.valid_dists(obs, tree.name_to_node, case['b'], case['f'])
            res.append(orc.runquery(tree, qn, obs, case['m']))
        _check(res, case['p'], '-d %s f=%s' % (case['m'], case['f']))
    stree = read_tree(os.path.join(DATA, 'small_backbone.nwk'))
    assert extended_newick(stree) == g['small_tree']
    for case in g['small']:
        res = []
        for qn, obs in read_dismat(os.path.join(DATA, 'small_dist.mat')):
            obs = orc.valid_dists(obs, stree.name_to_node, 25, 0.2)
            res.append(orc.runquery(stree, qn, obs, case['m']))
        _check(res, case['p'], 'small %s' % case['m'])
        assert res[0]['placements'][0]['p'][0][0] == 3


def test_g4_edge_cases(c1):
    tree, ref, qry = c1
    g = load_json('g4_placements.json')['edge_cases']
    L = ref.length
    allgap = np.full(L, ord('-'), dtype=np.uint8)
    seqs = {'allgap': allgap, ref.names[0]: ref.seqs[0], 'copy_of_second': ref.seqs[1], 'allgap2': allgap,
            'normal': qry.seqs[0]}
    assert list(seqs) == g['names']
    res = _run_aln(tree, ref, list(seqs), list(seqs.values()), 'OLS', 'MLSE', False, 0.2, 25)
    _check(res, g['results'], 'edge cases')
    joined = orc.join_jplace(res)
    assert [p['n'][0] for p in joined['placements']] == [w['n'] for w in g['joined']]
    # first result is kept although unplaceable; second all-gap one is dropped (jutil.py:11-18)
    assert joined['placements'][0]['p'][0][0] == -1 and 'allgap2' not in [p['n'][0] for p in joined['placements']]
    ex = load_json('g4_placements.json')['exclude_ME']
    res = _run_aln(tree, ref, qry.names, qry.seqs, 'OLS', 'ME', False, 0.2, 25, exclude=True)
    _check(res, ex, 'exclude')


def test_g5_tree_strings():
    import hashlib
    import json
    import re
    g = load_json('g5_tree_strings.json')
    s = json.load(open(os.path.join(DATA, 'prot', 'out.jplace')))['tree']
    assert hashlib.sha256(s.encode()).hexdigest() == g['prot_out_sha256']
    assert extended_newick(parse_newick(re.sub(r'\{\d+\}', '', s))) == s
    for name in ('small_backbone.nwk', 'backbone.nwk'):
        assert extended_newick(read_tree(os.path.join(DATA, name))) == g[name]
    nw = extended_newick(read_tree(os.path.join(DATA, 'prot', 'backbone.nwk')))
    assert hashlib.sha256(nw.encode()).hexdigest() == g['prot_backbone_sha256']


@pytest.mark.parametrize('label', ['nt_OLS', 'aa_FM'])
def test_g6_synthetic_alignment(label):
    g = load_json('g6_synthetic.json')[label]
    d = synth.make_dataset(g['N'], g['L'], g['Q'], protein=g['protein'])
    rows = {n: d.ref_seqs[i] for i, n in enumerate(d.ref_names)}
    reps = [(d.ref_seqs[i], [n]) for i, n in enumerate(d.ref_names)]
    fn = orc.scoredist if g['protein'] else orc.jc69
    ties = 0
    for i, qn in enumerate(d.query_names):
        obs = orc.get_obs_dist(d.query_seqs[i], reps, rows, fn, g['f'], g['b'], 0.001)
        assert len(obs) == g['n_obs'][i]
        r = orc.runquery(d.tree, qn, obs, g['m'])
        w = g['p'][i]
        bs = g['best_second'][i]
        if r['placements'][0]['p'][0][0] != w['p'][0] and bs is not None and \
                abs(bs[1] - bs[0]) <= 1e-12 * max(abs(bs[0]), 1e-300):
            ties += 1  # documented tie class: equal error to 12 digits
            continue
        assert_prow(r['placements'][0]['p'][0], w['p'], ctx='%s %s' % (label, qn))
    assert ties <= 2


@pytest.mark.parametrize('m', ['BME', 'OLS'])
def test_g6_synthetic_dmat(m):
    g = load_json('g6_synthetic.json')['dmat_' + m]
    d = synth.make_dataset(g['N'], 500, g['Q'])
    D = synth.noisy_distance_rows(d.tree, d.query_leaf, d.query_pendant, list(range(g['Q'])))
    for i, qn in enumerate(d.query_names):
        obs = orc.valid_dists(dict(zip(d.ref_names, D[i].tolist())), d.tree.name_to_node, g['b'], g['f'])
        r = orc.runquery(d.tree, qn, obs, m)
        assert_prow(r['placements'][0]['p'][0], g['p'][i]['p'], ctx='dmat %s %s' % (m, qn))


# ----------------------------------------------------------------------------- G10: -p with clusters
@pytest.fixture(scope='module')
def prot(tmp_path_factory):
    import prot_cases
    ref_fp, qry_fp, tree_fp = prot_cases.write_case(str(tmp_path_factory.mktemp('prot')))
    return read_tree(tree_fp), read_alignment(ref_fp, True, False), read_alignment(qry_fp, True, False)


def test_g10_protein_selection(prot):
    """get_obs_dist over scoredist with consensus representatives (apples/Reference.py:117-157): the observed dict in
    insertion order, singleton and clade clusters."""
    tree, ref, qry = prot
    g = load_json('g10_prot_clustered.json')
    assert extended_newick(tree) == g['tree']
    rows = {n: ref.seqs[i] for i, n in enumerate(ref.names)}
    reps_single = [(ref.seqs[i], [n]) for i, n in enumerate(ref.names)]
    reps_clades = _reps_from_json(g['clade_clusters'])
    for case in g['cases']:
        reps = reps_single if case['clusters'] == 'singleton' else reps_clades
        q = qry.seqs[qry.index[case['query']]]
        for fn in (orc.scoredist, orc.scoredist_sequential):
            obs = orc.get_obs_dist(q, reps, rows, fn, case['f'], case['b'], 0.001)
            assert list(obs) == [k for k, _ in case['obs']], case['query']
            np.testing.assert_allclose(list(obs.values()), [v for _, v in case['obs']], rtol=1e-12, atol=1e-15)
    assert any(v == 0 and np.signbit(v) for case in g['cases'] for _, v in case['obs'])  # the duplicate row's -0.0


def test_g10_protein_placements(prot):
    tree, ref, qry = prot
    g = load_json('g10_prot_clustered.json')
    rows = {n: ref.seqs[i] for i, n in enumerate(ref.names)}
    reps = _reps_from_json(g['clade_clusters'])
    for case in g['placements']:
        res = []
        for n, s in zip(qry.names, qry.seqs):
            obs = orc.get_obs_dist(s, reps, rows, orc.scoredist, case['f'], case['b'], 0.001)
            res.append(orc.runquery(tree, n, obs, case['m'], case['c'], case['n']))
        _check(res, case['p'], 'prot %s/%s/n=%s/f=%s' % (case['m'], case['c'], case['n'], case['f']))
