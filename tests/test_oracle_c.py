"""Pin the C oracle (oracle/oracle.c) against the reference's fixtures and the Python oracle. CPU only."""
import os
import sys

import numpy as np
import pytest

from helpers import DATA, GOLD, ROOT, assert_prow, load_json, read_dismat

sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import apples_oracle as orc  # noqa: E402
from oracle_c import COracle  # noqa: E402

from apples_amd import synth  # noqa: E402
from apples_amd.engine import jc69_lut, placement_row  # noqa: E402
from apples_amd.fasta import read_alignment  # noqa: E402
from apples_amd.tree import read_tree  # noqa: E402


@pytest.fixture(scope='module')
def c1():
    tree = read_tree(os.path.join(DATA, 'backbone.nwk'))
    ref = read_alignment(os.path.join(DATA, 'ref.fa'), False, False)
    qry = read_alignment(os.path.join(DATA, 'query.fa'), False, False)
    nodes = np.array([tree.name_to_node.get(n, -1) for n in ref.names], np.int32)
    return tree, ref, qry, nodes


def test_c_per_edge_bit_identical_to_reference(c1):
    tree, ref, qry, nodes = c1
    g = np.load(os.path.join(GOLD, 'g3_per_edge.npz'))
    co = COracle(tree)
    for qi in range(3):
        names = [str(x) for x in g['q%d_obs_names' % qi]]
        on = np.array([tree.name_to_node[k] for k in names], np.int32)
        od = g['q%d_obs_dist' % qi]
        for m in ('OLS', 'FM', 'BME', 'BE'):
            co.set_options(method=m)
            r = co.sweep_edges(on, od)
            key = 'q%d_%s_' % (qi, m)
            edges = g[key + 'edge']
            assert np.array_equal(np.nonzero(r['valid'])[0], edges) and r['lca'] == int(g[key + 'lca'])
            assert np.array_equal(r['S'][edges], g[key + 'S'])
            assert np.array_equal(r['R'][edges], g[key + 'R'])
            assert np.array_equal(r['x'][edges], g[key + 'x'])
            assert np.array_equal(r['err'][edges], g[key + 'err'])  # pow(x, 2.0) as CPython's x ** 2


def test_c_placements_match_golden(c1):
    tree, ref, qry, nodes = c1
    g = load_json('g4_placements.json')
    co = COracle(tree, ref.seqs, nodes, lut=jc69_lut(ref.length, 0.001))
    for case in g['aln']:
        if case.get('clusters') == 'clades':
            continue
        co.set_options(method=case['m'], criterion=case['c'], negative=case['n'], threshold=case['f'], baseobs=case['b'])
        for p, w in zip(co.place_sequences(qry.seqs), case['p']):
            assert_prow(placement_row(p), w['p'], ctx='C aln %s/%s %s' % (case['m'], case['c'], w['n']))
    rows = list(read_dismat(os.path.join(DATA, 'dist.mat')))
    cols = list(rows[0][1])
    D = np.array([[r[1][c] for c in cols] for r in rows])
    cn = np.array([tree.name_to_node.get(c, -1) for c in cols], np.int32)
    co = COracle(tree)
    for case in g['dist']:
        co.set_options(method=case['m'], threshold=case['f'], baseobs=case['b'])
        for p, w in zip(co.place_distances(D, cn), case['p']):
            assert_prow(placement_row(p), w['p'], ctx='C -d %s %s' % (case['m'], w['n']))


def test_c_edge_cases(c1):
    tree, ref, qry, nodes = c1
    g = load_json('g4_placements.json')['edge_cases']
    allgap = np.full(ref.length, ord('-'), np.uint8)
    names = g['names']
    seqs = np.vstack([allgap, ref.seqs[0], ref.seqs[1], allgap, qry.seqs[0]])
    self_rows = np.array([ref.index[n] if (n in ref.index and n in tree.name_to_node) else -1 for n in names], np.int32)
    co = COracle(tree, ref.seqs, nodes, method='OLS', lut=jc69_lut(ref.length, 0.001))
    for p, w in zip(co.place_sequences(seqs, self_rows), g['results']):
        assert_prow(placement_row(p), w['p'], ctx='C edge case %s' % w['n'])


def test_c_matches_python_oracle_on_synthetic():
    d = synth.make_dataset(800, 300, 48)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    rows = {n: d.ref_seqs[i] for i, n in enumerate(d.ref_names)}
    reps = [(d.ref_seqs[i], [n]) for i, n in enumerate(d.ref_names)]
    for m in ('OLS', 'FM', 'BME', 'BE'):
        for c in ('MLSE', 'ME', 'HYBRID'):
            co = COracle(d.tree, d.ref_seqs, nodes, method=m, criterion=c, lut=jc69_lut(300, 0.001), threads=2)
            out = co.place_sequences(d.query_seqs)
            for i, qn in enumerate(d.query_names):
                obs = orc.get_obs_dist(d.query_seqs[i], reps, rows, orc.jc69, 0.2, 25, 0.001)
                want = orc.runquery(d.tree, qn, obs, m, c)['placements'][0]['p'][0]
                got = placement_row(out[i])
                assert got == want, (m, c, qn, got, want)  # bit identical, types included
                assert out[i]['n_obs'] == len(obs)
