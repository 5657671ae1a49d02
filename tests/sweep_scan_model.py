"""Executable model of the scan formulation of the least-squares sweep (csrc/sweep_scan.hip).

TEST INFRASTRUCTURE: a plain-Python statement of the data layout and of the order of every
floating-point operation of the HIP kernel, checked against the oracle (tests/test_sweep_scan_model.py)
on CPU, where no GPU exists.  It is not used by the product.

Idea.  The observed leaves arrive sorted by node id (= left-to-right post-order number,
apples/util.py:57-69).  For leaf i let lev[i] be its level and lca[i] the level of the lowest common
ancestor of leaves i-1 and i.  The induced subtree (apples/Subtree.py:23-43) is then, with
top = min(lca[1:]) the level of the subtree's root:

    nodes = { ancestor of leaf i at level m : max(lca[i], top) < m <= lev[i] }   (leaf i "owns" them)

because an ancestor of leaf i at a level <= lca[i] is also an ancestor of leaf i-1 and is owned by an
earlier leaf.  Within one level the owned nodes, taken in leaf order, are sorted by node id, so

  * position of a node inside its level   = number of earlier leaves owning a node at that level
  * children of a node                     = a run of consecutive positions of the level below, in
                                             file order (post-order ids increase left to right)
  * parent of the node leaf i owns at m    = the node at level m-1 owned by the last leaf j <= i
                                             that owns one there.

Ballots and prefix counts over the leaves give every index; nothing is looked up in a hash table, a
bitmap or through tree pointers, and all per-level arrays are read and written sequentially.
Static per-tree tables: for every leaf the node id, edge length and smallest subtree id ("first") of
each of its ancestors, by level.
"""
import numpy as np


class LeafTables:
    """Per leaf, by level m = 0..lev: ancestor node id, its edge length, smallest node id of its subtree."""

    def __init__(self, tree):
        n = tree.n_nodes
        parent = np.asarray(tree.parent)
        self.level = np.asarray(tree.level)
        size = np.ones(n, dtype=np.int64)
        for v in range(n - 1):  # children before parents
            size[parent[v]] += size[v]
        first = np.arange(n) - size + 1
        self.off = {}
        self.node, self.e, self.first = [], [], []
        for x in np.nonzero(np.asarray(tree.is_leaf))[0]:
            chain = []
            v = int(x)
            while v >= 0:
                chain.append(v)
                v = int(parent[v])
            chain.reverse()  # index = level
            self.off[int(x)] = len(self.node)
            self.node += chain
            self.e += [float(tree.edge_len[v]) for v in chain]
            self.first += [int(first[v]) for v in chain]


def structure(tab, obs):
    """obs: observed leaf node ids, ascending.  Returns (top, levels) where levels[m] = list of entries
    dict(node, e, owner, parent_pos, is_leaf) for m = top+1 .. max level."""
    n = len(obs)
    lev = [int(tab.level[x]) for x in obs]
    lca = [0] * n
    for i in range(1, n):
        o = tab.off[obs[i]]
        # largest level whose ancestor of leaf i still contains leaf i-1 (first <= obs[i-1]); "first" is
        # non-decreasing with the level, so a binary search finds it (the kernel does that)
        m = 0
        while m + 1 <= lev[i] and tab.first[o + m + 1] <= obs[i - 1]:
            m += 1
        lca[i] = m
    top = min(lca[1:])
    lca[0] = top
    levels = {}
    for m in range(top + 1, max(lev) + 1):
        ent = []
        above = 0  # inclusive count of owners at level m-1 among leaves <= i
        for i in range(n):
            if m - 1 > top and lca[i] < m - 1 <= lev[i]:
                above += 1
            if lca[i] < m <= lev[i]:
                o = tab.off[obs[i]]
                ent.append(dict(node=tab.node[o + m], e=tab.e[o + m], owner=i, is_leaf=(lev[i] == m),
                                parent_pos=(above - 1) if m - 1 > top else -1))
        levels[m] = ent
    return top, levels


def sweep(tab, obs, dist, method, lift, leaf_tuple):
    """S and R tuples of every node of the induced subtree, in the kernel's order of operations.
    Returns ({node: S}, {node: R}, lca node)."""
    top, levels = structure(tab, obs)
    bme = method == 'BME'
    ms = sorted(levels)
    # ---- bottom-up: level m+1 -> m, position-wise over the children
    for m in ms:
        for en in levels[m]:
            en['S'] = leaf_tuple(method, dist[en['owner']]) if en['is_leaf'] else None
            en['kid0'], en['nk'] = -1, 0
    for m in reversed(ms):
        kids = levels[m]
        c = 0
        while c < len(kids):
            P = kids[c]['parent_pos']
            ln = 1
            while c + ln < len(kids) and kids[c + ln]['parent_pos'] == P:
                ln += 1
            if P >= 0:
                coef = 1 / ln if bme else None
                acc = [0, 0, 0, 0, 0, 0]
                for k in range(ln):
                    t = lift(kids[c + k]['S'], kids[c + k]['e'])
                    for x in range(6):
                        acc[x] += t[x] if coef is None else coef * t[x]
                par = levels[m - 1][P]
                assert not par['is_leaf'] and par['S'] is None
                par['S'], par['kid0'], par['nk'] = tuple(acc), c, ln
            c += ln
    # ---- top-down: children of level m-1 at level m, position-wise over the children
    for m in ms:
        kids = levels[m]
        for c, en in enumerate(kids):
            P = en['parent_pos']
            if P >= 0:
                par = levels[m - 1][P]
                kid0, nk = par['kid0'], par['nk']
            else:
                par = None
                kid0, nk = 0, len(kids)
            coef = 1 / ((1 if par is not None else 0) + nk - 1) if bme else None
            acc = [0, 0, 0, 0, 0, 0]
            for s in range(kid0, kid0 + nk):
                if s == c:
                    continue
                t = lift(kids[s]['S'], kids[s]['e'])
                for x in range(6):
                    acc[x] += t[x] if coef is None else coef * t[x]
            if par is not None:
                t = lift(par['R'], par['e'])
                for x in range(6):
                    acc[x] += t[x] if coef is None else coef * t[x]
            en['R'] = tuple(acc)
    S = {en['node']: en['S'] for m in ms for en in levels[m]}
    R = {en['node']: en['R'] for m in ms for en in levels[m]}
    o = tab.off[obs[0]]
    return S, R, tab.node[o + top]
