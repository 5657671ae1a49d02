"""oracle/fasttree_me.py against the outputs of the FastTree binary the reference bundles (tests/golden/g9_*,
made by tests/golden/make_goldens.py g9 from apples/tools/FastTree-linux run as reestimateBackbone.py:82-84 runs it)."""
import os
import sys

import numpy as np
import pytest

from helpers import ROOT

sys.path.insert(0, ROOT)
from apples_amd import reestimate as R  # noqa: E402
from apples_amd.fasta import read_records  # noqa: E402
from oracle import fasttree_me  # noqa: E402
from fasttree_cases import fasttree_case, fasttree_cases  # noqa: E402

GOLD = os.path.dirname(os.path.abspath(__file__)) + '/golden'
PRINT_RES = 5.5e-6  # FastTree prints five decimals (half a unit = 5e-6) of sums it forms in single precision (lengths up to 3.0
# in the saturated sets: a few 1e-7 on top)


def splits(root):
    """{leaf set on the side without the smallest label: summed length}; a two-child root's edges are one split."""
    allv = frozenset(x.label for x in root.leaves())
    ref = min(allv)
    out = {}
    below = {}
    order, st = [], [root]
    while st:
        v = st.pop()
        order.append(v)
        st.extend(v.children)
    for v in reversed(order):
        s = frozenset([v.label]) if not v.children else frozenset().union(*[below[id(c)] for c in v.children])
        below[id(v)] = s
        if v.parent is not None:
            key = s if ref not in s else allv - s
            out[key] = out.get(key, 0.0) + (v.length or 0.0)
    return out


def oracle_tree(newick, seq_of, protein):
    """The input topology with the oracle's lengths (a two-child root: the one edge's length on the first child)."""
    root = R.from_newick(newick)
    nodes, parent, children = R.flatten(root)
    leaf_seq = [None if children[v] else seq_of[nodes[v].label] for v in range(len(nodes))]
    got = fasttree_me.branch_lengths(len(nodes), parent, children, leaf_seq, protein)
    for v, nd in enumerate(nodes):
        nd.length = float(got[v]) if parent[v] >= 0 else None
    if len(root.children) == 2:
        root.children[1].length = 0.0
    return root


def check(newick, seq_of, protein, gold_name):
    want = splits(R.from_newick(open(os.path.join(GOLD, gold_name)).read()))
    got = splits(oracle_tree(newick, seq_of, protein))
    assert set(want) == set(got)
    worst = max(abs(want[k] - got[k]) for k in want)
    assert worst <= PRINT_RES, worst
    return worst


def test_oracle_matches_fasttree_on_the_reference_test_data():
    data = os.path.join(GOLD, 'data')
    with open(os.path.join(data, 'ref.fa')) as f:
        seq_of = {n: np.frombuffer(s.encode(), np.uint8) for n, s in read_records(f)}
    check(open(os.path.join(data, 'backbone.nwk')).read(), seq_of, False, 'g9_fasttree_data.nwk')


@pytest.mark.parametrize('name', sorted(fasttree_cases()))
def test_oracle_matches_fasttree_on_rooted_inputs(name):
    n, L, protein, seed, odd, mean_len = fasttree_cases()[name]
    d, seqs = fasttree_case(n, L, protein, seed, odd, mean_len)
    check(d.newick, dict(zip(d.ref_names, seqs)), protein, 'g9_fasttree_%s.nwk' % name)


def test_no_shared_site_is_distance_three():
    # two leaves with disjoint site coverage next to each other: FastTree's distance for "no overlap" is 3.0
    seqs = [np.frombuffer(s, np.uint8) for s in (b'ACGT----', b'----ACGT', b'ACGTACGT', b'ACGAACGA')]
    children = [[], [], [], [], [0, 1], [4, 2, 3]]
    parent = [4, 4, 5, 5, 5, -1]
    out = fasttree_me.branch_lengths(6, parent, children, seqs + [None, None], False)
    # leaf 0: (d(0,1) + d(0,up) - d(1,up)) / 2 with d(0,1) = 3
    assert out[0] + out[1] == pytest.approx(3.0, abs=1e-12)
