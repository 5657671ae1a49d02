"""Backbone tree as flat arrays (the layout the HIP kernels read).

Replaces the treeswift objects the reference walks (apples/prepareTree.py:24-36)
with struct-of-arrays indexed by ``edge_index``:

* node id == ``edge_index`` == 0-based left-to-right post-order number
  (apples/util.py:57-69), so children < parent and the root is ``n_nodes - 1``;
* ``level`` == BFS depth, root 0 (apples/util.py:72-88);
* children in file order as CSR (``child_off``/``child_idx``) -- the order the
  reference accumulates S/R sums in (apples/OLS.py:36, 59).

Newick reader follows the contract in SURVEY.md Appendix B: ``[&R]`` prefix ->
rooted, labels may be quoted, ``[...]`` comments dropped, lengths via float().
"""
import re

import numpy as np

_TOKEN = re.compile(r"\(|\)|,|;|:|'(?:[^']|'')*'|\[[^\]]*\]|[^(),:;\[\]']+")


class Tree:
    """Array form of the backbone tree.  All arrays have length ``n_nodes``."""

    def __init__(self, parent, edge_len, has_len, labels, child_off, child_idx, level, is_rooted, label_src=None):
        self.parent = parent  # int32, -1 for the root
        self.edge_len = edge_len  # float64, 0.0 where has_len is False
        self.has_len = has_len  # bool: edge length present in the Newick
        # labels: list[str|None], or None with label_src = (text, offset[n], length[n]: -1 = no label) -- the labels as slices of the
        # Newick text they were scanned from, cut into strings only when somebody asks (a 200 000-leaf backbone: 0.12 s of 0.35)
        self._labels = labels
        self._label_src = label_src
        self.child_off = child_off  # int32[n_nodes+1]
        self.child_idx = child_idx  # int32[n_nodes-1], file order
        self.level = level  # int32
        self.is_rooted = is_rooted
        self.n_nodes = len(parent)
        self.root = self.n_nodes - 1
        nchild = np.diff(child_off)
        self.is_leaf = nchild == 0
        self.leaves = np.nonzero(self.is_leaf)[0].astype(np.int32)  # ascending id == left-to-right
        # leaf label -> node id (apples/prepareTree.py:32-34; later duplicates overwrite)
        if labels is None:
            text, lo, ll = label_src
            a, k = lo[self.leaves], ll[self.leaves]
            if (k < 0).any():  # (a leaf without a label: the key is None, as labels[v] would be)
                self.name_to_node = {(text[o:o + n] if n >= 0 else None): v
                                     for o, n, v in zip(a.tolist(), k.tolist(), self.leaves.tolist())}
            else:
                self.name_to_node = dict(zip([text[o:e] for o, e in zip(a.tolist(), (a + k).tolist())], self.leaves.tolist()))
        else:
            self.name_to_node = {}
            for v in self.leaves:
                self.name_to_node[labels[v]] = int(v)

    @property
    def labels(self):
        """list[str|None], one per node."""
        if self._labels is None:
            text, lo, ll = self._label_src
            self._labels = [text[o:o + k] if k >= 0 else None for o, k in zip(lo.tolist(), ll.tolist())]
        return self._labels

    @property
    def n_leaves(self):
        return len(self.leaves)

    def children(self, v):
        return self.child_idx[self.child_off[v]:self.child_off[v + 1]]


def _scan_py(text):
    """Token loop of the reader.  Returns, per node in creation (= pre-) order: parent (creation
    index), depth, subtree size, label (str or None), branch length (nan where absent) and whether
    one was given."""
    # creation-order (= pre-order) temporary nodes; depth and subtree size come out of the token
    # loop, so the post-order numbering needs no traversal
    t_parent = [-1]
    t_label = [None]
    t_len = [None]
    t_depth = [0]
    t_size = {}  # internal nodes only: nodes in the subtree, known when the ')' closes it
    cur = 0
    depth = 0
    expect_len = False
    done = False
    for m in _TOKEN.finditer(text):
        tok = m.group(0)
        c = tok[0]
        if expect_len:
            if c in '(),;' or c == '[':
                raise ValueError('malformed Newick: missing branch length before %r' % tok)
            t_len[cur] = float(tok.strip())
            expect_len = False
            continue
        if c == '(':
            depth += 1
            t_parent.append(cur)
            t_label.append(None)
            t_len.append(None)
            t_depth.append(depth)
            cur = len(t_parent) - 1
        elif c == ',':
            par = t_parent[cur]
            if par < 0:
                raise ValueError('malformed Newick: comma at top level')
            t_parent.append(par)
            t_label.append(None)
            t_len.append(None)
            t_depth.append(depth)
            cur = len(t_parent) - 1
        elif c == ')':
            cur = t_parent[cur]
            if cur < 0:
                raise ValueError('malformed Newick: unbalanced parentheses')
            depth -= 1
            t_size[cur] = len(t_parent) - cur
        elif c == ':':
            expect_len = True
        elif c == ';':
            done = True
            break
        elif c == '[':
            continue  # comment
        elif c == "'":
            t_label[cur] = tok[1:-1].replace("''", "'")
        else:
            s = tok.strip()
            if s:
                t_label[cur] = s
    if not done and cur != 0:
        raise ValueError('malformed Newick: unexpected end of input')
    if cur != 0:
        raise ValueError('malformed Newick: unbalanced parentheses')
    n = len(t_parent)
    size = np.ones(n, dtype=np.int64)
    if t_size:
        size[np.fromiter(t_size.keys(), dtype=np.int64, count=len(t_size))] = \
            np.fromiter(t_size.values(), dtype=np.int64, count=len(t_size))
    lens = np.array([np.nan if x is None else x for x in t_len], dtype=np.float64)
    given = np.array([x is not None for x in t_len], dtype=bool)
    return np.array(t_parent, dtype=np.int64), np.array(t_depth, dtype=np.int64), size, t_label, lens, given


class _LabelSlices:
    """Per-node labels in creation order as (offset, length) into the scanned text (length -1 = none)."""
    def __init__(self, text, off, length):
        self.text, self.off, self.length = text, off, length


def _scan_native(text):
    """The same through libapples_io.so (include/apples_io.h: apples_newick_scan), or None when the
    library is missing, the text is not plain ASCII, or the scanner met something it leaves to
    :func:`_scan_py` (malformed input: that parser words the error)."""
    from .fasta import _load_io
    lib = _load_io()
    if lib is None or not hasattr(lib, 'apples_newick_scan') or not text.isascii():
        return None
    import ctypes
    raw = text.encode('ascii')
    cap = 1 + raw.count(b'(') + raw.count(b',')
    par = np.empty(cap, np.int32); dep = np.empty(cap, np.int32); siz = np.empty(cap, np.int32)
    loff = np.empty(cap, np.int64); llen = np.empty(cap, np.int32); lq = np.empty(cap, np.uint8)
    val = np.empty(cap, np.float64); st = np.empty(cap, np.uint8)
    voff = np.empty(cap, np.int64); vlen = np.empty(cap, np.int32)
    n = ctypes.c_int64(0)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.apples_newick_scan(raw, len(raw), cap, ptr(par), ptr(dep), ptr(siz), ptr(loff), ptr(llen), ptr(lq),
                                ptr(val), ptr(st), ptr(voff), ptr(vlen), ctypes.byref(n))
    if rc != 0:
        return None
    n = n.value
    # labels stay slices of the text (Tree.labels cuts them on demand) unless a quoted one holds a doubled quote to undo
    quoted = np.nonzero(lq[:n])[0].tolist()
    if any("''" in text[loff[k]:loff[k] + llen[k]] for k in quoted):
        lo = loff[:n].tolist(); ll = llen[:n].tolist()
        labels = [text[o:o + k] if k >= 0 else None for o, k in zip(lo, ll)]
        for k in quoted:
            labels[k] = labels[k].replace("''", "'")
    else:
        labels = _LabelSlices(text, loff[:n].copy(), llen[:n].copy())
    st = st[:n]
    lens = np.where(st == 1, val[:n], np.nan)
    try:
        for k in np.nonzero(st == 2)[0].tolist():  # spellings left to float(): nan, inf, 1_000, ...
            lens[k] = float(text[voff[k]:voff[k] + vlen[k]])
    except ValueError:
        return None  # not a number: _scan_py reports the first one in file order
    return par[:n].astype(np.int64), dep[:n].astype(np.int64), siz[:n].astype(np.int64), labels, lens, st != 0


def parse_newick(text):
    """Parse one Newick tree string into a :class:`Tree`."""
    text = text.strip()
    is_rooted = False
    if text.startswith('[&R]'):
        is_rooted = True
        text = text[4:].lstrip()
    elif text.startswith('[&U]'):
        text = text[4:].lstrip()

    scanned = _scan_native(text)
    if scanned is None:
        scanned = _scan_py(text)
    pre_parent, pre_depth, size, t_label, lens, given = scanned
    n = len(pre_parent)
    # left-to-right post-order number (apples/util.py:65-69) of the node created k-th (pre-order):
    # k - depth + size - 1  (the nodes before it in pre-order that are not its ancestors, plus its
    # own descendants)
    post = np.arange(n, dtype=np.int64) - pre_depth + size - 1

    parent = np.full(n, -1, dtype=np.int32)
    nonroot = pre_parent >= 0
    parent[post[nonroot]] = post[pre_parent[nonroot]]
    edge_len = np.zeros(n, dtype=np.float64)
    has_len = np.zeros(n, dtype=bool)
    edge_len[post[given]] = lens[given]
    has_len[post] = given
    label_src = None
    if isinstance(t_label, _LabelSlices):
        lo = np.empty(n, dtype=np.int64); ll = np.empty(n, dtype=np.int32)
        lo[post] = t_label.off; ll[post] = t_label.length
        labels, label_src = None, (t_label.text, lo, ll)
    else:
        labels = [None] * n
        for p, lab in zip(post.tolist(), t_label):
            labels[p] = lab
    nchild = np.bincount(parent[parent >= 0], minlength=n)
    child_off = np.zeros(n + 1, dtype=np.int32)
    child_off[1:] = np.cumsum(nchild)
    # children in file order = creation order: sort the non-root nodes by (parent's number, creation order)
    kids = np.nonzero(nonroot)[0]
    order = kids[np.argsort(post[pre_parent[kids]], kind='stable')]
    child_idx = post[order].astype(np.int32) if n > 1 else np.empty(0, dtype=np.int32)
    level = np.zeros(n, dtype=np.int32)
    level[post] = pre_depth  # BFS depth (apples/util.py:72-88)
    return Tree(parent, edge_len, has_len, labels, child_off, child_idx, level, is_rooted, label_src=label_src)


def read_tree(path):
    with open(path) as f:
        return parse_newick(f.read())


def _len_str(x):
    # apples/jutil.py:80-87 -- integral floats print as ints, others as str(float)
    if float(x).is_integer():
        return str(int(x))
    return str(float(x))


def _extended_newick_native(tree):
    """The same through libapples_io.so (apples_extended_newick), or None: library missing, labels that are not plain strings."""
    from .fasta import _load_io
    lib = _load_io()
    if lib is None or not hasattr(lib, 'apples_extended_newick'):
        return None
    import ctypes
    n = tree.n_nodes
    src = getattr(tree, '_label_src', None)
    if src is not None and getattr(tree, '_labels', None) is None:
        # the labels are slices of the (ASCII) text the tree was scanned from: that text is the blob
        blob = src[0].encode('ascii')
        loff = np.ascontiguousarray(src[1], dtype=np.int64)
        llen = np.ascontiguousarray(src[2], dtype=np.int32)
        cap = 64 + 40 * n + int(np.maximum(llen, 0).sum())
    else:
        labs = tree.labels
        try:
            parts = [b'' if x is None else x.encode() for x in labs]
        except AttributeError:
            return None
        llen = np.fromiter((-1 if x is None else len(b) for x, b in zip(labs, parts)), dtype=np.int32, count=n)
        loff = np.zeros(n, dtype=np.int64)
        np.cumsum(np.maximum(llen[:-1], 0), out=loff[1:])
        blob = b''.join(parts)
        cap = 64 + 40 * n + len(blob)
    out = ctypes.create_string_buffer(cap)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.apples_extended_newick.restype = ctypes.c_int64
    lib.apples_extended_newick.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p,
                                           ctypes.c_int64]
    el = np.ascontiguousarray(tree.edge_len, dtype=np.float64)
    hl = np.ascontiguousarray(tree.has_len, dtype=np.uint8)
    co = np.ascontiguousarray(tree.child_off, dtype=np.int32)
    ci = np.ascontiguousarray(tree.child_idx, dtype=np.int32)
    k = lib.apples_extended_newick(n, ptr(co), ptr(ci), int(tree.root), ptr(el), ptr(hl), blob, ptr(loff), ptr(llen), out, cap)
    if k < 0:
        return None
    s = out.raw[:k].decode()
    return ('[&R] %s;' % s) if tree.is_rooted else ('%s;' % s)


def extended_newick(tree):
    """Newick with ``{edge_index}`` after every non-root node (apples/jutil.py:22-96)."""
    s = _extended_newick_native(tree)
    if s is not None:
        return s
    return _extended_newick_py(tree)


def _extended_newick_py(tree):
    n = tree.n_nodes
    off = tree.child_off.tolist()
    idx = tree.child_idx.tolist()
    labels = tree.labels
    # what follows a node's own text inside its parent's parentheses: ':length' (if any) and '{edge_index}'
    suffix = [(':%s{%d}' % (_len_str(x), c)) if h else '{%d}' % c
              for c, (x, h) in enumerate(zip(tree.edge_len.tolist(), tree.has_len.tolist()))]
    strs = [None] * n
    for v in range(n):  # ascending id is a post-order
        lab = labels[v]
        a, b = off[v], off[v + 1]
        if a == b:
            strs[v] = '' if lab is None else str(lab)
        else:
            inner = ','.join([strs[c] + suffix[c] for c in idx[a:b]])
            for c in idx[a:b]:
                strs[c] = None
            strs[v] = '(' + inner + ')' if lab is None else '(' + inner + ')' + str(lab)
    s = strs[tree.root]
    if tree.is_rooted:
        return '[&R] %s;' % s
    return '%s;' % s
