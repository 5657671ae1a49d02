"""Seeded synthetic inputs: backbone tree, simulated alignment, queries.

Deterministic from (seed, N, L, Q, alphabet).  Used by bench.py, the parity
tests and the golden generator; nothing here is on the placement hot path.

* Tree: N leaves ``t0..``, random-join binary topology (repeatedly join two
  random roots), branch lengths Exp(mean 0.01).
* Alignment: uniform root sequence evolved down every edge with per-site
  substitution probability 3/4(1-exp(-4t/3)) (nt, JC69) or 19/20(1-exp(-20t/19))
  (aa, Poisson), substitutions uniform over the other letters; then 5 % of the
  sites of every sequence -> '-'.
* Queries: random leaf edge, attached at its midpoint, pendant Exp(0.01).
"""
import numpy as np

from .tree import parse_newick

NT = np.frombuffer(b'ACGT', dtype=np.uint8)
AA = np.frombuffer(b'ARNDCQEGHILKMFPSTWYV', dtype=np.uint8)


def random_tree_newick(n_leaves, seed=1, mean_len=0.01, spine=0):
    """Random-join binary tree as a Newick string (iterative, O(N)).  spine > 0: the joining stops at spine + 1
    subtrees, which are hung on a caterpillar of `spine` nodes (the deep-backbone shape: more than `spine` levels)."""
    rng = np.random.default_rng(seed)
    # node table: children of internal nodes; leaves 0..n-1
    left = []
    right = []
    roots = list(range(n_leaves))
    nxt = n_leaves
    picks = rng.random(2 * n_leaves)
    pi = 0
    while len(roots) > 1 + max(int(spine), 0):
        i = int(picks[pi] * len(roots)); pi += 1
        roots[i], roots[-1] = roots[-1], roots[i]
        a = roots.pop()
        j = int(picks[pi] * len(roots)); pi += 1
        roots[j], roots[-1] = roots[-1], roots[j]
        b = roots.pop()
        left.append(a)
        right.append(b)
        roots.append(nxt)
        nxt += 1
    while len(roots) > 1:  # the caterpillar: (subtree, (subtree, (... (subtree, subtree))))
        b = roots.pop()
        a = roots.pop()
        left.append(a)
        right.append(b)
        roots.append(nxt)
        nxt += 1
    total = nxt
    lens = rng.exponential(mean_len, size=total)
    # iterative Newick emission
    out = []
    stack = [(roots[0], 0)]
    while stack:
        v, st = stack.pop()
        if v < n_leaves:
            out.append('t%d:%.6f' % (v, lens[v]))
            continue
        k = v - n_leaves
        if st == 0:
            out.append('(')
            stack.append((v, 1))
            stack.append((left[k], 0))
        elif st == 1:
            out.append(',')
            stack.append((v, 2))
            stack.append((right[k], 0))
        else:
            if v == roots[0]:
                out.append(')')
            else:
                out.append('):%.6f' % lens[v])
    out.append(';')
    return ''.join(out)


def reshape_newick(tree, kind, seed=11, frac=0.01):
    """The same backbone with another shape, as Newick text (same leaves in the same left-to-right order, same root
    distances of the leaves): what real inputs look like beside the strictly binary random-join tree --
    * 'unrooted': the root's first internal child is dissolved into the root (a root trifurcation: what FastTree prints and
      what the reference's own example backbones have);
    * 'polytomies': a fraction `frac` of the internal non-root nodes is dissolved into their parents (seeded)."""
    n = tree.n_nodes
    kids = [list(map(int, tree.children(v))) for v in range(n)]
    lens = np.array(tree.edge_len, dtype=np.float64)
    labels = tree.labels
    root = tree.root
    gone = np.zeros(n, bool)
    if kind == 'unrooted':
        for c in kids[root]:
            if kids[c]:
                gone[c] = True
                break
    elif kind == 'polytomies':
        rng = np.random.default_rng(seed)
        internal = np.array([v for v in range(n) if kids[v] and v != root])
        gone[internal[rng.random(len(internal)) < frac]] = True
    else:
        raise ValueError(kind)
    out = []
    stack = [(root, 0, 0.0)]  # node, next child, length inherited from dissolved ancestors
    while stack:
        v, k, add = stack.pop()
        if not kids[v]:
            out.append('%s:%.6f' % (labels[v], lens[v] + add))
            continue
        if gone[v]:  # its children take its place (and its length)
            for c in reversed(kids[v]):
                stack.append((c, 0, add + lens[v]))
            continue
        if k == 0:
            out.append('(')
            stack.append((v, 1, add))
            # the children in file order, a marker between them
            seq = []
            for c in kids[v]:
                seq.append((c, 0, 0.0))
            for item in reversed(seq):
                stack.append(item)
        else:
            out.append(')' if v == root else '):%.6f' % (lens[v] + add))
    # commas: between consecutive siblings = wherever a token that ends a subtree is followed by one that starts one
    text = []
    for i, tok in enumerate(out):
        if i and out[i - 1] != '(' and not tok.startswith(')'):
            text.append(',')
        text.append(tok)
    text.append(';')
    return ''.join(text)


def reshape_tree(tree, kind, seed=11, frac=0.01):
    return parse_newick(reshape_newick(tree, kind, seed, frac))


def _sub_prob(t, n_states):
    k = n_states / (n_states - 1.0)
    return (1.0 - 1.0 / n_states) * (1.0 - np.exp(-k * t))


def _evolve(idx, t, n_states, rng):
    """idx: int8 state array; returns a mutated copy after time t."""
    p = _sub_prob(t, n_states)
    hit = rng.random(idx.shape[0]) < p
    nh = int(hit.sum())
    out = idx.copy()
    if nh:
        out[hit] = (idx[hit] + rng.integers(1, n_states, size=nh)) % n_states
    return out


class SynthData:
    pass


def make_dataset(n_leaves, length, n_queries, protein=False, seed_tree=1, seed_aln=5, seed_query=3,
                 gap_rate=0.05, mean_len=0.01, spine=0):
    """Returns an object with: newick, tree, ref_names, ref_seqs uint8[N,L] (rows in
    leaf order), query_names, query_seqs uint8[Q,L], query_leaf (true sister leaf)."""
    alphabet = AA if protein else NT
    ns = len(alphabet)
    newick = random_tree_newick(n_leaves, seed_tree, mean_len, spine)
    tree = parse_newick(newick)
    rng = np.random.default_rng(seed_aln)
    n = tree.n_nodes
    states = np.empty((n, length), dtype=np.int8)
    states[tree.root] = rng.integers(0, ns, size=length)
    for v in range(n - 2, -1, -1):  # parents (larger ids) first
        states[v] = _evolve(states[tree.parent[v]], tree.edge_len[v], ns, rng)
    leaves = tree.leaves
    ref = alphabet[states[leaves]]
    gaps = rng.random(ref.shape) < gap_rate
    ref[gaps] = ord('-')
    ref_names = [tree.labels[v] for v in leaves]

    rq = np.random.default_rng(seed_query)
    q_leaf = rq.integers(0, len(leaves), size=n_queries)
    q_pend = rq.exponential(mean_len, size=n_queries)
    qs = np.empty((n_queries, length), dtype=np.uint8)
    for i in range(n_queries):
        v = leaves[q_leaf[i]]
        mid = _evolve(states[tree.parent[v]], tree.edge_len[v] / 2.0, ns, rq)
        s = _evolve(mid, q_pend[i], ns, rq)
        row = alphabet[s]
        row[rq.random(length) < gap_rate] = ord('-')
        qs[i] = row
    d = SynthData()
    d.newick = newick
    d.tree = tree
    d.ref_names = ref_names
    d.ref_seqs = np.ascontiguousarray(ref)
    d.query_names = ['q%d' % i for i in range(n_queries)]
    d.query_seqs = qs
    d.query_leaf = q_leaf
    d.query_pendant = q_pend
    return d


def noisy_distance_rows(tree, query_leaf, query_pendant, rows, seed_noise=7, rel_sd=0.05, floor=1e-4):
    """``-d`` style input (C5): for the queries in ``rows``, true path distance from the
    attachment point to every leaf x (1 + rel_sd N(0,1)), floored.  float64[len(rows), n_leaves],
    columns in leaf order."""
    n = tree.n_nodes
    leaves = tree.leaves
    out = np.empty((len(rows), len(leaves)), dtype=np.float64)
    for k, qi in enumerate(rows):
        rng = np.random.default_rng([seed_noise, int(qi)])
        v = int(leaves[query_leaf[qi]])
        # distance from the midpoint of v's edge to every node: walk up, then sweep down
        dist = np.full(n, -1.0)
        half = tree.edge_len[v] / 2.0
        dist[v] = half
        up = half
        u = v
        while tree.parent[u] >= 0:
            u2 = int(tree.parent[u])
            up = up + (tree.edge_len[u] if u != v else 0.0)
            dist[u2] = up
            u = u2
        for w in range(n - 2, -1, -1):
            if dist[w] < 0:
                dist[w] = dist[tree.parent[w]] + tree.edge_len[w]
        true = dist[leaves] + query_pendant[qi]
        noisy = true * (1.0 + rel_sd * rng.standard_normal(len(leaves)))
        out[k] = np.maximum(noisy, floor)
    return out


class TreeIndex:
    """Root distances and leaf-rank ranges of every node (leaves in id order are contiguous per
    subtree because node ids are post-order numbers)."""

    def __init__(self, tree):
        n = tree.n_nodes
        self.rd = np.zeros(n)
        for v in range(n - 2, -1, -1):
            self.rd[v] = self.rd[tree.parent[v]] + tree.edge_len[v]
        cnt = tree.is_leaf.astype(np.int64)
        for v in range(n - 1):
            cnt[tree.parent[v]] += cnt[v]
        rank_end = np.cumsum(tree.is_leaf)  # leaves with id <= v
        self.last = rank_end           # exclusive end of v's leaf-rank range
        self.first = rank_end - cnt    # inclusive start
        self.leaf_rd = self.rd[tree.leaves]


def fast_distance_rows(tree, index, query_leaf, query_pendant, rows, seed_noise=7, rel_sd=0.05, floor=1e-4):
    """Vectorised form of :func:`noisy_distance_rows` for benchmark-size inputs (C5): path distance
    from the attachment point (midpoint of the sister leaf's edge) to every leaf via
    rd(x) + rd(l) - 2 rd(lca), the lca being constant on O(depth) contiguous leaf-rank ranges.
    Different noise stream than the per-query generator; not used for golden fixtures."""
    leaves = tree.leaves
    out = np.empty((len(rows), len(leaves)))
    rng = np.random.default_rng([seed_noise, 12345])
    for k, qi in enumerate(rows):
        v = int(leaves[query_leaf[qi]])
        half = tree.edge_len[v] / 2.0
        rdx = index.rd[v] - half            # root distance of the attachment point
        chain = []
        u = int(tree.parent[v])
        while u >= 0:
            chain.append(u)
            u = int(tree.parent[u])
        lca_rd = np.empty(len(leaves))
        for a in reversed(chain):           # root first; deeper ancestors overwrite their sub-ranges
            lca_rd[index.first[a]:index.last[a]] = index.rd[a]
        true = rdx + index.leaf_rd - 2.0 * lca_rd
        true[index.first[v]] = half         # the sister leaf itself
        true = true + query_pendant[qi]
        out[k] = np.maximum(true * (1.0 + rel_sd * rng.standard_normal(len(leaves))), floor)
    return out
