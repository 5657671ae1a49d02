"""apples_backbone_lengths (csrc/backbone_me.hip) through the C ABI: against the oracle on the same inputs and
against the outputs of the FastTree binary the reference bundles (tests/golden/g9_*; reestimateBackbone.py:82-84)."""
import os
import sys
import time

import numpy as np
import pytest

from helpers import DATA, ROOT

sys.path.insert(0, ROOT)
from apples_amd import engine, synth  # noqa: E402
from apples_amd import reestimate as R  # noqa: E402
from apples_amd.fasta import read_records  # noqa: E402
from oracle import fasttree_me  # noqa: E402
from fasttree_cases import fasttree_case, fasttree_cases  # noqa: E402
from test_fasttree_me import GOLD, PRINT_RES, splits  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-9  # double arithmetic on both sides; only the order of the site sums differs


def both(newick, seq_of, protein, site_chunk=0):
    root = R.from_newick(newick)
    nodes, parent, children = R.flatten(root)
    leaves = [v for v in range(len(nodes)) if not children[v]]
    rows = np.stack([seq_of[nodes[v].label] for v in leaves])
    leaf_row = [-1] * len(nodes)
    for i, v in enumerate(leaves):
        leaf_row[v] = i
    got = engine.backbone_lengths(parent, children, leaf_row, rows, protein, 0, site_chunk)
    want = fasttree_me.branch_lengths(len(nodes), parent, children, [rows[r] if r >= 0 else None for r in leaf_row], protein)
    return root, nodes, parent, got, want


def against_fixture(root, nodes, parent, got, gold_name):
    for v, nd in enumerate(nodes):
        nd.length = float(got[v]) if parent[v] >= 0 else None
    if len(root.children) == 2:
        root.children[1].length = 0.0
    want = splits(R.from_newick(open(os.path.join(GOLD, gold_name)).read()))
    have = splits(root)
    assert set(want) == set(have)
    assert max(abs(want[k] - have[k]) for k in want) <= PRINT_RES


def test_reference_test_data_unrooted():
    with open(os.path.join(DATA, 'ref.fa')) as f:
        seq_of = {n: np.frombuffer(s.encode(), np.uint8) for n, s in read_records(f)}
    root, nodes, parent, got, want = both(open(os.path.join(DATA, 'backbone.nwk')).read(), seq_of, False)
    assert np.abs(got - want).max() <= TOL
    against_fixture(root, nodes, parent, got, 'g9_fasttree_data.nwk')


@pytest.mark.parametrize('name', sorted(fasttree_cases()))
def test_rooted_inputs_nucleotide_with_odd_symbols_and_protein(name):
    n, L, protein, seed, odd, mean_len = fasttree_cases()[name]
    d, seqs = fasttree_case(n, L, protein, seed, odd, mean_len)
    seq_of = dict(zip(d.ref_names, seqs))
    root, nodes, parent, got, want = both(d.newick, seq_of, protein)
    assert np.abs(got - want).max() <= TOL
    against_fixture(root, nodes, parent, got, 'g9_fasttree_%s.nwk' % name)
    # the same walked in 64-site chunks: sums of the same terms
    got64 = both(d.newick, seq_of, protein, site_chunk=64)[3]
    assert np.abs(got64 - got).max() <= 1e-12


def test_root_shapes_and_no_overlap():
    seqs = {k: np.frombuffer(v, np.uint8) for k, v in
            dict(a=b'ACGT----ACGTAAAA', b=b'----ACGTACGTAAAT', c=b'ACGTACGTACGTAATT', d=b'ACGAACGAACGTATTT', e=b'ACGAACTAACTTTTTT').items()}
    for nw in ('((a,b),c,(d,e));', '(a,((b,c),(d,e)));', '((a,b),(c,(d,e)));', '(((a,b),c),d,e);'):
        root, nodes, parent, got, want = both(nw, seqs, False)
        assert np.abs(got - want).max() <= TOL, nw
    # a and b share no site: their distance is FastTree's 3.0
    root, nodes, parent, got, want = both('((a,b),c,(d,e));', {**seqs, 'a': np.frombuffer(b'ACGT------------', np.uint8),
                                                                 'b': np.frombuffer(b'----ACGTACGTAAAT', np.uint8)}, False)
    la = {nd.label: got[v] for v, nd in enumerate(nodes) if nd.label}
    assert la['a'] + la['b'] == pytest.approx(3.0, abs=1e-12)


def test_refusals():
    seqs = {k: np.frombuffer(b'ACGTACGT', np.uint8) for k in 'abcde'}
    with pytest.raises(RuntimeError, match='resolve polytomies'):
        both('((a,b,c),d,e);', seqs, False)
    with pytest.raises(RuntimeError, match='resolve polytomies'):
        both('(a,b,c,d,e);', seqs, False)
    root = R.from_newick('((a,b),c,(d,e));')
    nodes, parent, children = R.flatten(root)
    rows = np.stack([seqs[k] for k in 'abcde'])
    with pytest.raises(RuntimeError, match='no alignment row'):
        engine.backbone_lengths(parent, children, [-1] * len(nodes), rows, False)
    with pytest.raises(RuntimeError, match='multiple of 64'):
        leaf_row = [-1] * len(nodes)
        for i, v in enumerate(v for v in range(len(nodes)) if not children[v]):
            leaf_row[v] = i
        engine.backbone_lengths(parent, children, leaf_row, rows, False, 0, 100)


def test_larger_backbone_timing():
    """20 000 leaves x 1 000 sites: every branch against the oracle on a subsample is too slow on the CPU, so:
    finite, the chunked walk equals the one-pass walk, and the time is printed."""
    d = synth.make_dataset(20000, 1000, 1, seed_tree=11)
    root = R.from_newick(d.newick)
    nodes, parent, children = R.flatten(root)
    leaves = [v for v in range(len(nodes)) if not children[v]]
    row_of = {n: i for i, n in enumerate(d.ref_names)}
    leaf_row = [-1] * len(nodes)
    for v in leaves:
        leaf_row[v] = row_of[nodes[v].label]
    engine.backbone_lengths(parent, children, leaf_row, d.ref_seqs, False)
    t0 = time.time()
    got = engine.backbone_lengths(parent, children, leaf_row, d.ref_seqs, False)
    dt = time.time() - t0
    print('backbone lengths, 20000 leaves x 1000 sites: %.1f ms' % (dt * 1e3))
    assert np.isfinite(got).all()
    part = engine.backbone_lengths(parent, children, leaf_row, d.ref_seqs, False, 0, 256)
    assert np.abs(part - got).max() <= 1e-12
    # true branch lengths of the simulation are recovered on average (sanity of scale, not parity)
    true = np.array([nd.length or 0.0 for nd in nodes])
    assert abs(got.sum() / true.sum() - 1) < 0.1
