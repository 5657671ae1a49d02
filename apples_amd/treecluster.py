"""Max-diameter tree clustering for the reduced reference.

The reference shells out to the third-party tool ``TreeCluster.py -m max -t 1.2*f``
(apples/Reference.py:87-88) and reads its ``name<TAB>cluster`` table.  That tool is not part of
this build (and not vendored in the reference), so its published linear-time algorithm is
restated here from the TreeCluster paper (Balaban, Moshiri, Mai, Jia, Mirarab 2019, "max"
method: the fewest clusters such that every pairwise leaf distance inside a cluster is <= t):

* polytomies are resolved into zero-length binary nodes, missing lengths count as 0;
* post-order sweep; every node keeps the largest distance to a not-yet-clustered leaf through
  its left and through its right child; when left + right > t the child on the longer side is cut
  off and its remaining leaves become a cluster;
* what is left at the end is one more cluster; one-leaf clusters get the id ``-1``.

The sweep also exists natively (include/apples_io.h: apples_max_clusters), statement for statement;
:func:`max_clusters` uses it when libapples_io.so is there, and the two are compared in the tests.

**Parity unpinned**: no TreeCluster run exists in this environment to compare against; the
polytomy resolution order and cluster numbering follow the tool's documented behaviour as
recalled, not a verified trace.  Cluster assignment is an *input* of the hot path (a
``--clusters`` file overrides this module), so the hot path's parity does not depend on it.
"""
from collections import deque


def _binarize(tree):
    """children lists of a binary-resolved copy; new nodes get ids >= n_nodes and edge length 0.
    Resolution order as treeswift's resolve_polytomies: repeatedly replace the last two children
    by a new zero-length parent of them."""
    import numpy as np
    n = tree.n_nodes
    off = tree.child_off.tolist()
    idx = tree.child_idx.tolist()
    children = [idx[off[v]:off[v + 1]] for v in range(n)]
    elen = np.where(tree.has_len, tree.edge_len, 0.0).tolist()
    if n == 0 or int(np.max(np.diff(tree.child_off))) <= 2:
        return children, elen  # already binary
    q = deque([tree.root])
    while q:
        v = q.popleft()
        ch = children[v]
        while len(ch) > 2:
            c1 = ch.pop()
            c2 = ch.pop()
            nid = len(children)
            children.append([c1, c2])
            elen.append(0.0)
            ch.append(nid)
        q.extend(ch)
    return children, elen


def _postorder(children, root, n_original):
    if len(children) == n_original:  # no node was added: ids are post-order numbers already (apples/util.py:65-69)
        return range(n_original)
    order = []
    stack = [(root, 0)]
    while stack:
        v, i = stack.pop()
        ch = children[v]
        if i < len(ch):
            stack.append((v, i + 1))
            stack.append((ch[i], 0))
        else:
            order.append(v)
    return order


def _max_clusters_native(tree, threshold):
    """The sweep through libapples_io.so (include/apples_io.h: apples_max_clusters); None without the library."""
    import ctypes
    import numpy as np
    from .fasta import _load_io
    lib = _load_io()
    if lib is None or not hasattr(lib, 'apples_max_clusters') or tree.n_nodes == 0:
        return None
    off = np.ascontiguousarray(tree.child_off, np.int32)
    idx = np.ascontiguousarray(tree.child_idx, np.int32)
    elen = np.ascontiguousarray(np.where(tree.has_len, tree.edge_len, 0.0), np.float64)
    n_leaves = int(np.count_nonzero(np.diff(off) == 0))
    order = np.empty(max(n_leaves, 1), np.int32)
    ends = np.empty(n_leaves + 2, np.int32)
    n_cl = ctypes.c_int32(0)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    if lib.apples_max_clusters(tree.n_nodes, ptr(off), ptr(idx), ptr(elen), int(tree.root), float(threshold), ptr(order),
                               ptr(ends), ctypes.byref(n_cl)) != 0:
        return None
    labels = tree.labels
    names = [labels[u] for u in order[:int(ends[n_cl.value])].tolist()]
    e = ends[:n_cl.value + 1].tolist()
    return [names[a:b] for a, b in zip(e[:-1], e[1:])]


def max_clusters(tree, threshold):
    """[[leaf label, ...], ...] in the order the sweep closes them (the remainder last)."""
    native = _max_clusters_native(tree, threshold)
    if native is not None:
        return native
    return _max_clusters_py(tree, threshold)


def _max_clusters_py(tree, threshold):
    children, elen = _binarize(tree)
    total = len(children)
    deleted = [False] * total
    left = [0.0] * total
    right = [0.0] * total
    clusters = []

    def cut(v):
        out = []
        st = [v]
        while st:
            u = st.pop()
            if deleted[u]:
                continue
            deleted[u] = True
            if not children[u]:
                out.append(tree.labels[u])
            st.extend(reversed(children[u]))
        return out

    for v in _postorder(children, tree.root, tree.n_nodes):
        if deleted[v]:
            continue
        ch = children[v]
        if not ch:
            left[v] = right[v] = 0.0
            continue
        if len(ch) == 1:  # unifurcation: pass the child's depth through
            c = ch[0]
            left[v] = 0.0 if deleted[c] else max(left[c], right[c]) + elen[c]
            right[v] = 0.0
            continue
        a, b = ch
        if deleted[a] and deleted[b]:
            cut(v)
            continue
        left[v] = 0.0 if deleted[a] else max(left[a], right[a]) + elen[a]
        right[v] = 0.0 if deleted[b] else max(left[b], right[b]) + elen[b]
        if left[v] + right[v] > threshold:
            if left[v] > right[v]:
                cluster = cut(a)
                left[v] = 0.0
            else:
                cluster = cut(b)
                right[v] = 0.0
            if cluster:
                clusters.append(cluster)
    rest = cut(tree.root)
    if rest:
        clusters.append(rest)
    return clusters


def cluster_table(tree, threshold):
    """[(leaf label, cluster id string)] as the tool's output table lists them: clusters in the
    order found, numbered from 1; one-leaf clusters are ``-1``."""
    rows = []
    num = 1
    for cl in max_clusters(tree, threshold):
        if len(cl) == 1:
            rows.append((cl[0], '-1'))
        else:
            rows.extend((name, str(num)) for name in cl)
            num += 1
    return rows


def write_table(tree, threshold, path):
    with open(path, 'w') as f:
        f.write('SequenceName\tClusterNumber\n')
        for name, cid in cluster_table(tree, threshold):
            f.write('%s\t%s\n' % (name, cid))


def grouped(tree, threshold):
    """[(cluster id, [names])] as apples/Reference.py:93-100 groups the table (stable sort by id as
    a string, then groupby)."""
    import itertools
    rows = sorted(cluster_table(tree, threshold), key=lambda x: x[1])
    return [(k, [r[0] for r in g]) for k, g in itertools.groupby(rows, lambda x: x[1])]
