"""The scan formulation of the sweep (tests/sweep_scan_model.py, the CPU model of csrc/sweep_scan.hip)
against the oracle's pointer-walking restatement of apples/Subtree.py + all_S_values/all_R_values:
same induced subtree, same LCA, S and R tuples bit for bit, on the reference's own trees (binary with
a trifurcating root; polytomies up to degree 42 with zero and negative branch lengths)."""
import os
import sys

import numpy as np
import pytest

from helpers import DATA, ROOT

sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import apples_oracle as orc  # noqa: E402
from apples_amd.tree import read_tree  # noqa: E402
from sweep_scan_model import LeafTables, sweep  # noqa: E402


@pytest.mark.parametrize('tree_file', ['backbone.nwk', os.path.join('prot', 'backbone.nwk'), 'small_backbone.nwk'])
def test_scan_formulation_equals_pointer_walk(tree_file):
    tree = read_tree(os.path.join(DATA, tree_file))
    tab = LeafTables(tree)
    leaves = np.nonzero(np.asarray(tree.is_leaf))[0]
    rng = np.random.default_rng(5)
    sizes = [2, 3, 4, 7, 25, 60, 200, len(leaves)]
    for trial, k in enumerate(sizes * 2):
        k = min(k, len(leaves))
        if trial % 3 == 2 and k < len(leaves):  # a clade-like set: consecutive leaves plus a few outliers
            a = int(rng.integers(0, len(leaves) - k + 1))
            obs = sorted(set(leaves[a:a + k].tolist()) | set(rng.choice(leaves, size=min(3, len(leaves)), replace=False).tolist()))
        else:
            obs = sorted(rng.choice(leaves, size=k, replace=False).tolist())
        dist = rng.uniform(0.01, 1.2, size=len(obs)).tolist()
        leaf_dist = dict(zip(obs, dist))
        valid, lca, num = orc.induced_subtree(tree, obs)
        for method in ('OLS', 'FM', 'BME', 'BE'):
            want_S = orc.s_values(tree, valid, leaf_dist, method)
            want_R = orc.r_values(tree, valid, lca, want_S, method)
            S, R, got_lca = sweep(tab, obs, dist, method, orc._LIFT[method], orc._leaf_tuple)
            assert got_lca == lca
            assert sorted(S) == sorted(want_S) and len(S) == num
            for v in want_S:
                assert tuple(map(float, S[v])) == tuple(map(float, want_S[v])), (method, v, 'S')
                assert tuple(map(float, R[v])) == tuple(map(float, want_R[v])), (method, v, 'R')
                # (same bits, signed zeros included)
                assert np.array_equal(np.array(S[v], float).view(np.int64), np.array(want_S[v], float).view(np.int64))
                assert np.array_equal(np.array(R[v], float).view(np.int64), np.array(want_R[v], float).view(np.int64))
