"""Backbone branch re-estimation wrapper (apples_amd/reestimate.py; apples/reestimateBackbone.py:22-118):
polytomy resolution, Newick round trip, re-rooting FastTree's unrooted answer on the input's root edge.
FastTree itself is external: a stand-in script here, the reference's bundled binary when this container
has it."""
import os
import stat
import sys
import types

import numpy as np
import pytest

from helpers import DATA, ROOT

sys.path.insert(0, ROOT)
from apples_amd import reestimate as R  # noqa: E402
from apples_amd import synth  # noqa: E402
from apples_amd.tree import parse_newick  # noqa: E402


def _splits(root):
    """{frozenset(leaf labels below an edge): edge length} for every non-root node."""
    out = {}

    def walk(v):
        if not v.children:
            s = frozenset([v.label])
        else:
            s = frozenset().union(*[walk(c) for c in v.children])
        if v.parent is not None:
            out[s] = v.length
        return s
    sys.setrecursionlimit(100000)
    walk(root)
    return out


def test_resolve_polytomies_and_unifurcations():
    t = R.from_newick('((A:1,B:2,C:3,D:4):1,((E:1):2):3,F:1,G:1,H:1);')
    t = R.suppress_unifurcations(t)
    assert 'E' in [c.label for c in t.children] and [c.length for c in t.children if c.label == 'E'] == [6.0]
    assert len(t.children) == 5
    R.resolve_polytomies(t)
    stack = [t]
    while stack:
        v = stack.pop()
        assert len(v.children) in (0, 2)
        stack.extend(v.children)
    assert sorted(x.label for x in t.leaves()) == list('ABCDEFGH')
    # new internal nodes sit on zero-length edges; original edges keep their lengths
    s = _splits(t)
    assert s[frozenset('ABCD')] == 1.0 and s[frozenset('A')] == 1.0 and s[frozenset('D')] == 4.0
    assert parse_newick(R.to_newick(t)).n_leaves == 8


def _unroot(root):
    """What FastTree prints for a rooted binary tree: the root's first internal child is merged into it."""
    a, b = root.children
    keep, merge = (a, b) if b.children else (b, a)
    root.children = [keep] + merge.children
    for c in merge.children:
        c.parent = root
    keep.length = keep.length + merge.length
    return root


@pytest.mark.parametrize('seed', [1, 2, 3, 4])
def test_rerooting_restores_the_input_root(seed):
    nw = synth.random_tree_newick(40, seed=seed)
    orig = R.from_newick(nw)
    want = _splits(orig)
    left, right = orig.children
    if left.children:
        two, one, l2, l1 = [c.first_leaf().label for c in left.children], right.first_leaf().label, left.length, right.length
    else:
        two, one, l2, l1 = [c.first_leaf().label for c in right.children], left.first_leaf().label, right.length, left.length
    ft = _unroot(R.from_newick(nw))
    assert len(ft.children) == 3
    by = {x.label: x for x in ft.leaves()}
    ft = R.reroot(ft, by[one], None)
    m = R.mrca(ft, [by[x] for x in two])
    ml = m.length
    ft = R.reroot(ft, m, ml / 2)
    for i in range(2):
        if ft.children[i] is m:
            ft.children[i].length = ml * l2 / (l2 + l1)
            ft.children[1 - i].length = ml * l1 / (l2 + l1)
    got = _splits(ft)
    assert set(got) == set(want) and len(ft.children) == 2
    for k in want:
        assert got[k] == pytest.approx(want[k], rel=1e-12, abs=1e-15)


def _options(tree_fp, ref_fp, exe=None, protein=False):
    return types.SimpleNamespace(tree_fp=tree_fp, ref_fp=ref_fp, protein_seqs=protein, fasttree_fp=exe)


def test_wrapper_with_a_stand_in_fasttree(tmp_path, monkeypatch):
    stub = tmp_path / 'FastTree'
    with open(stub, 'w') as f:
        f.write('#!%s\nimport sys\nsys.path.insert(0, %r); sys.path.insert(0, %r)\n'
                'from apples_amd import reestimate as R\nfrom test_reestimate import _unroot\n'
                'a = sys.argv\nassert "-nosupport" in a and "-nome" in a and "-noml" in a and "-nt" in a\n'
                'assert sys.stdin.read(1) == ">"\n'
                't = R.from_newick(open(a[a.index("-intree") + 1]).read())\n'
                'stack = [t]\n'
                'while stack:\n    v = stack.pop(); stack.extend(v.children)\n    v.length = None if v.length is None else 2 * v.length\n'
                'print(R.to_newick(_unroot(t)))\n' % (sys.executable, ROOT, os.path.join(ROOT, 'tests')))
    os.chmod(stub, os.stat(stub).st_mode | stat.S_IEXEC)
    nw = synth.random_tree_newick(30, seed=7)
    tree_fp = tmp_path / 'bb.nwk'
    open(tree_fp, 'w').write(nw + '\n')
    ref_fp = tmp_path / 'ref.fa'
    open(ref_fp, 'w').write('>t0\nACGT\n')
    monkeypatch.setenv('PATH', str(tmp_path / 'nowhere'))
    monkeypatch.delenv('APPLES_FASTTREE', raising=False)
    o = _options(str(tree_fp), str(ref_fp))
    assert R.find_fasttree(None) is None and R.find_fasttree('native') is None  # this build's own estimator
    with pytest.raises(ValueError):
        R.find_fasttree(str(tmp_path / 'missing'))
    # found through the environment
    monkeypatch.setenv('APPLES_FASTTREE', str(stub))
    assert R.reestimate_backbone(o) is True and o.tree_fp != str(tree_fp)
    want = _splits(R.from_newick(nw))
    got_root = R.from_newick(open(o.tree_fp).read())
    got = _splits(got_root)
    assert set(got) == set(want) and len(got_root.children) == 2
    for k in want:
        assert got[k] == pytest.approx(2 * want[k], rel=1e-12)


BUNDLED = '/root/reference/apples/tools/FastTree-linux'


@pytest.mark.skipif(not os.path.exists(BUNDLED), reason='the reference tree (with its bundled FastTree) is only in the build container')
def test_wrapper_with_the_bundled_fasttree_on_the_example_data():
    """End to end on data/backbone.nwk + data/ref.fa with the FastTree the reference ships: same leaf set,
    same splits (the topology is fixed by -intree), same root edge, finite lengths, and the result is a
    tree this build's reader and the placement path accept."""
    o = _options(os.path.join(DATA, 'backbone.nwk'), os.path.join(DATA, 'ref.fa'), BUNDLED)
    assert R.reestimate_backbone(o) is True
    before = R.from_newick(open(os.path.join(DATA, 'backbone.nwk')).read())
    after = R.from_newick(open(o.tree_fp).read())
    assert sorted(x.label for x in before.leaves()) == sorted(x.label for x in after.leaves())
    sb, sa = _splits(before), _splits(after)
    # data/backbone.nwk has a trifurcating root (unrooted): FastTree's answer is used as it comes
    assert len(before.children) == 3
    allb = frozenset(x.label for x in before.leaves())
    canon = lambda s: min(s, allb - s, key=lambda x: (len(x), sorted(x)))  # noqa: E731  (a split, whichever side is below)
    assert {canon(s) for s in sb} == {canon(s) for s in sa}
    assert all(v is not None and np.isfinite(v) for v in sa.values())
    t = parse_newick(open(o.tree_fp).read())
    assert t.n_leaves == 490


@pytest.mark.skipif(not os.path.exists(BUNDLED), reason='the reference tree (with its bundled FastTree) is only in the build container')
def test_bundled_fasttree_on_a_rooted_tree_keeps_the_root_edge(tmp_path):
    d = synth.make_dataset(60, 400, 1)
    tree_fp, ref_fp = tmp_path / 'bb.nwk', tmp_path / 'ref.fa'
    open(tree_fp, 'w').write(d.newick + '\n')
    with open(ref_fp, 'w') as f:
        for n, s in zip(d.ref_names, d.ref_seqs):
            f.write('>%s\n%s\n' % (n, s.tobytes().decode()))
    o = _options(str(tree_fp), str(ref_fp), BUNDLED)
    assert R.reestimate_backbone(o) is True
    before, after = R.from_newick(d.newick), R.from_newick(open(o.tree_fp).read())
    assert len(after.children) == 2
    assert set(_splits(before)) == set(_splits(after))  # rooted splits: the root edge is where it was
    lb = [c.length for c in before.children]
    la = {frozenset(x.label for x in c.leaves()): c.length for c in after.children}
    kb = {frozenset(x.label for x in c.leaves()): c.length for c in before.children}
    tot_a, tot_b = sum(la.values()), sum(lb)
    for k in kb:  # the new root-edge length is split in the input's proportion (reestimateBackbone.py:103-110)
        assert la[k] / tot_a == pytest.approx(kb[k] / tot_b, rel=1e-9)


def test_native_route_of_the_wrapper(tmp_path, monkeypatch):
    """No FastTree executable: the lengths come from apples_backbone_lengths (here replaced by the oracle, the
    GPU is not needed for the host logic): topology, child order and rooting stay, the root edge gets ONE
    estimate split in the input's proportion, lengths carry five decimals, polytomies are resolved first."""
    from apples_amd import engine
    from oracle import fasttree_me

    def on_cpu(parent, children, leaf_row, rows, protein, device=0, site_chunk=0):
        seqs = [rows[r] if r >= 0 else None for r in leaf_row]
        return fasttree_me.branch_lengths(len(parent), parent, children, seqs, protein)

    monkeypatch.setattr(engine, 'backbone_lengths', on_cpu)
    monkeypatch.setenv('PATH', str(tmp_path / 'nowhere'))
    monkeypatch.delenv('APPLES_FASTTREE', raising=False)
    d = synth.make_dataset(40, 300, 1)
    nw = d.newick
    tree_fp, ref_fp = tmp_path / 'bb.nwk', tmp_path / 'ref.fa'
    open(tree_fp, 'w').write(nw + '\n')
    with open(ref_fp, 'w') as f:
        for n, s in zip(d.ref_names, d.ref_seqs):
            f.write('>%s\n%s\n' % (n, s.tobytes().decode()))
        f.write('>not_in_tree\n%s\n' % ('A' * 300))
    o = _options(str(tree_fp), str(ref_fp))
    assert R.reestimate_backbone(o) is True
    before, after = R.from_newick(nw), R.from_newick(open(o.tree_fp).read())
    assert [x.label for x in before.leaves()] == [x.label for x in after.leaves()]  # order kept
    assert set(_splits(before)) == set(_splits(after)) and len(after.children) == 2
    lb, la = [c.length for c in before.children], [c.length for c in after.children]
    assert la[0] / sum(la) == pytest.approx(lb[0] / sum(lb), rel=1e-9)
    assert round(sum(la), 5) == pytest.approx(sum(la), abs=1e-12)
    stack = [c for r in after.children for c in r.children]
    while stack:
        v = stack.pop()
        stack.extend(v.children)
        assert float('%.5f' % v.length) == v.length
    R.cleanup(o)
    # a leaf without a sequence is an error, as it is for FastTree
    open(tree_fp, 'w').write(nw.replace('t3:', 'nobody:') + '\n')
    o = _options(str(tree_fp), str(ref_fp))
    with pytest.raises(ValueError, match='no sequence'):
        R.reestimate_backbone(o)
    # polytomy: resolved into zero-length joins before the estimate (reestimateBackbone.py:40-46)
    star = '((a:0.1,b:0.1,c:0.1,d:0.1):0.1,e:0.2,f:0.2);'
    open(tree_fp, 'w').write(star + '\n')
    with open(ref_fp, 'w') as f:
        for n, s in zip('abcdef', ('ACGTACGTAC', 'ACGTACGTAA', 'ACGTACGAAA', 'ACGTAAGAAA', 'TCGTAAGAAT', 'TCTTAAGAAT')):
            f.write('>%s\n%s\n' % (n, s))
    o = _options(str(tree_fp), str(ref_fp))
    assert R.reestimate_backbone(o) is True
    after = R.from_newick(open(o.tree_fp).read())
    assert len(after.children) == 3 and sorted(x.label for x in after.leaves()) == list('abcdef')
    assert all(len(v.children) in (0, 2) for c in after.children for v in [c])


# what the bundled binary (apples/tools/FastTree-linux -nosupport -nome -noml -intree ... -nt) printed for two-child roots
# (topology and child order only; recorded in the build container)
FASTTREE_UNROOTING = [
    ('((A,B),(C,(D,E)));', '((C,(D,E)),A,B);'),
    ('((C,(D,E)),(A,B));', '((A,B),C,(D,E));'),
    ('(A,(B,(C,(D,E))));', '(A,B,(C,(D,E)));'),
    ('((B,(C,(D,E))),A);', '(A,B,(C,(D,E)));'),
    ('(((A,B),F),(C,(D,E)));', '((C,(D,E)),(A,B),F);'),
    ('((C,(D,E)),((A,B),F));', '(((A,B),F),C,(D,E));'),
]


@pytest.mark.parametrize('rooted,unrooted', FASTTREE_UNROOTING)
def test_unrooting_follows_the_bundled_fasttree(rooted, unrooted):
    root = R.from_newick(rooted)
    root.children[0].length = 0.25  # the one estimate of the root edge (native_lengths puts it on the first child)
    root.children[1].length = 0.0
    R.unroot_like_fasttree(root)
    import re
    assert re.sub(r':[0-9.e-]+', '', R.to_newick(root)) == unrooted
    kept = [c for c in root.children if c.length == 0.25]
    assert len(root.children) == 3 and len(kept) == 1  # the surviving side carries the root edge's length


def test_native_route_without_all_lengths_prints_fasttrees_unrooted_answer(tmp_path, monkeypatch):
    """A rooted input that lacks a branch length: the reference does not re-root (reestimateBackbone.py:49,91) and
    places on FastTree's unrooted answer; so does the native route (node count and edge numbering follow)."""
    from apples_amd import engine
    from oracle import fasttree_me

    def on_cpu(parent, children, leaf_row, rows, protein, device=0, site_chunk=0):
        return fasttree_me.branch_lengths(len(parent), parent, children, [rows[r] if r >= 0 else None for r in leaf_row], protein)

    monkeypatch.setattr(engine, 'backbone_lengths', on_cpu)
    monkeypatch.setenv('PATH', str(tmp_path / 'nowhere'))
    monkeypatch.delenv('APPLES_FASTTREE', raising=False)
    tree_fp, ref_fp = tmp_path / 'bb.nwk', tmp_path / 'ref.fa'
    open(tree_fp, 'w').write('((A:0.1,B:0.1):0.1,(C:0.1,(D,E:0.1):0.1):0.1);\n')  # D has no length
    with open(ref_fp, 'w') as f:
        for n, s in zip('ABCDE', ('ACGTACGTACGTACGTAAAA', 'ACGTACGTACGAACGTAAAT', 'ACGTTCGTACGAACGTATAT', 'ACGATCGTACGAACCTATAT',
                                  'ACGATCGAACGAACCTATTT')):
            f.write('>%s\n%s\n' % (n, s))
    o = _options(str(tree_fp), str(ref_fp))
    assert R.reestimate_backbone(o) is True
    after = R.from_newick(open(o.tree_fp).read())
    import re
    assert re.sub(r':-?[0-9.e-]+', '', R.to_newick(after)) == '((C,(D,E)),A,B);'
    # lengths: the bundled binary printed ((C:-0.02420,(D:-0.01798,E:0.12531):0.14153):0.14153,A:0.12531,B:-0.01798);
    want = {'A': 0.12531, 'B': -0.01798, 'C': -0.02420, 'D': -0.01798, 'E': 0.12531}
    for leaf in after.leaves():
        assert leaf.length == pytest.approx(want[leaf.label], abs=6e-6)
    assert after.children[0].length == pytest.approx(0.14153, abs=6e-6)
    R.cleanup(o)
