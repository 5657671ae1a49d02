"""Backbone branch-length re-estimation before placement (apples/reestimateBackbone.py:22-118).

The reference resolves polytomies, hands the topology to FastTree (``-nosupport -nome -noml -intree``,
reestimateBackbone.py:82-84: branch lengths by FastTree's distance-based minimum-evolution estimates, JC69 or
BLOSUM45 corrected), re-roots FastTree's unrooted answer where the input was rooted (:91-111) and places
on that tree.  It ships FastTree as a prebuilt third-party binary; this build does not.  An executable is
looked up -- ``--fasttree``, ``$APPLES_FASTTREE``, then ``FastTree``, ``FastTree-linux``, ``fasttree`` on PATH
-- and used exactly as the reference uses it; when there is none (or with ``--fasttree native``) the lengths
come from this build's own estimator on the GPU (``apples_backbone_lengths``, csrc/backbone_me.hip: FastTree's
profile-based balanced minimum-evolution lengths, equal to the binary's output to its five printed decimals
-- tests/golden/g9_*).  The native path keeps the input topology, child order and rooting as they are: the
root edge gets the one estimated length split in the input's proportion (:103-110), and identical sequences
are not collapsed (FastTree joins them under zero-length branches, rearranging the tree).

Everything around the estimator is this module's own: the mutable tree, ``suppress_unifurcations``,
``resolve_polytomies`` (treeswift's scheme as remembered: the last two children are joined under a new
zero-length node until two remain -- parity unpinned, treeswift is absent here as it is in the survey),
Newick out, re-rooting on the edge that separated the input root's two sides with the new length split in
the input's proportion.
"""
import logging
import os
import shutil
import subprocess
import sys
import tempfile
import time
from collections import deque

from .tree import parse_newick


class Node:
    __slots__ = ('label', 'length', 'children', 'parent')

    def __init__(self, label=None, length=None):
        self.label = label
        self.length = length
        self.children = []
        self.parent = None

    def add(self, child):
        child.parent = self
        self.children.append(child)

    def leaves(self):
        out, stack = [], [self]
        while stack:
            v = stack.pop()
            if v.children:
                stack.extend(reversed(v.children))
            else:
                out.append(v)
        return out

    def first_leaf(self):
        v = self
        while v.children:
            v = v.children[0]
        return v


def from_newick(text):
    """Mutable tree from Newick text (through this build's reader, apples_amd.tree.parse_newick)."""
    t = parse_newick(text)
    nodes = [Node(t.labels[v], float(t.edge_len[v]) if t.has_len[v] else None) for v in range(t.n_nodes)]
    for v in range(t.n_nodes):
        for c in t.children(v):
            nodes[v].add(nodes[int(c)])
    return nodes[t.root]


def _fmt(x):
    # repr keeps every digit; integers print without a fraction like the jplace tree string does
    return str(int(x)) if float(x).is_integer() and abs(x) < 1e15 else repr(float(x))


def _label(s):
    if s is None:
        return ''
    if any(c in s for c in "()[]':;, \t\n"):
        return "'" + s.replace("'", "''") + "'"
    return s


def to_newick(root):
    out = []
    stack = [(root, 0)]
    while stack:
        v, k = stack.pop()
        if not v.children:
            out.append(_label(v.label))
        elif k == 0:
            out.append('(')
            stack.append((v, 1))
            stack.append((v.children[0], 0))
            continue
        elif k < len(v.children):
            out.append(',')
            stack.append((v, k + 1))
            stack.append((v.children[k], 0))
            continue
        else:
            out.append(')' + _label(v.label))
        if v.parent is not None and v.length is not None:
            out.append(':' + _fmt(v.length))
    return ''.join(out) + ';'


def suppress_unifurcations(root):
    """Nodes with a single child disappear; the child takes over the summed edge length."""
    stack = [root]
    order = []
    while stack:
        v = stack.pop()
        order.append(v)
        stack.extend(v.children)
    for v in reversed(order):
        if len(v.children) == 1 and v.parent is not None:
            c = v.children[0]
            if v.length is not None or c.length is not None:
                c.length = (c.length or 0.0) + (v.length or 0.0)
            p = v.parent
            p.children[p.children.index(v)] = c
            c.parent = p
    while len(root.children) == 1:  # a unary root hands the root over to its child
        root = root.children[0]
        root.parent = None
        root.length = None
    return root


def resolve_polytomies(node):
    """Below `node`: while a node has more than two children, its last two are joined under a new node on a
    zero-length edge (breadth first)."""
    q = deque([node])
    while q:
        v = q.popleft()
        while len(v.children) > 2:
            c1 = v.children.pop()
            c2 = v.children.pop()
            nn = Node(None, 0)
            v.add(nn)
            nn.add(c1)
            nn.add(c2)
        q.extend(v.children)


def reroot(root, node, length):
    """New root on the edge above `node`, `length` up from it (length = None or 0: `node`'s parent end is
    kept whole and the new root sits at the node's parent side with a zero split).  Returns the new root."""
    if node.parent is None:
        return root
    total = node.length
    up = length if length else 0.0
    new_root = Node(None, None)
    # walk from node's parent to the old root reversing the edges
    p = node.parent
    p.children.remove(node)
    node.parent = None
    new_root.add(node)
    node.length = up if total is not None else None
    carry = (total - up) if total is not None else None
    prev = new_root
    cur = p
    while cur is not None:
        nxt = cur.parent
        nxt_len = cur.length
        if nxt is not None:
            nxt.children.remove(cur)
        cur.parent = None
        prev.add(cur)
        cur.length = carry
        carry = nxt_len
        prev = cur
        cur = nxt
    # the old root may be left with one child (it was binary): suppress it
    return suppress_unifurcations(new_root)


def mrca(root, leaves):
    """Lowest node whose subtree holds every leaf in `leaves` (node objects), in the tree as rooted now."""
    want = set(id(x) for x in leaves)
    count = {}
    best = None
    stack = [(root, 0)]
    while stack:
        v, k = stack.pop()
        if k == 0:
            stack.append((v, 1))
            for c in v.children:
                stack.append((c, 0))
        else:
            c = (1 if id(v) in want else 0) + sum(count[id(x)] for x in v.children)
            count[id(v)] = c
            if c == len(want) and best is None:
                best = v
    return best


def flatten(root):
    """(nodes, parent, children): nodes in preorder, parent[v] (-1 at the root), children[v] lists in file order."""
    nodes, index, st = [], {}, [root]
    while st:
        v = st.pop()
        index[id(v)] = len(nodes)
        nodes.append(v)
        st.extend(reversed(v.children))
    parent = [index[id(v.parent)] if v.parent is not None else -1 for v in nodes]
    children = [[index[id(c)] for c in v.children] for v in nodes]
    return nodes, parent, children


def find_fasttree(explicit=None):
    """Path of the FastTree executable to run, or None for this build's own estimator."""
    if explicit == 'native' or (not explicit and os.environ.get('APPLES_FASTTREE') == 'native'):
        return None
    for cand in (explicit, os.environ.get('APPLES_FASTTREE')):
        if cand:
            if os.path.isfile(cand) and os.access(cand, os.X_OK):
                return cand
            raise ValueError('FastTree executable %s not found or not executable' % cand)
    for name in ('FastTree', 'FastTree-linux', 'fasttree'):
        p = shutil.which(name)
        if p:
            return p
    return None


def _read_rows(ref_fp, labels):
    """Byte matrix [len(labels), L] of the FASTA's own characters for the given sequence names."""
    import numpy as np
    from .fasta import read_records
    want = {x: i for i, x in enumerate(labels)}
    rows, length = [None] * len(labels), None
    with open(ref_fp) as f:
        for name, seq in read_records(f):
            i = want.get(name)
            if i is None:
                continue
            if length is None:
                length = len(seq)
            elif len(seq) != length:
                raise ValueError('Sequence %s has length %d, the alignment has %d columns' % (name, len(seq), length))
            rows[i] = np.frombuffer(seq.encode('latin-1'), dtype=np.uint8)
    missing = [x for x, r in zip(labels, rows) if r is None]
    if missing:
        raise ValueError('%d backbone leaves have no sequence in %s (first: %s)' % (len(missing), ref_fp, missing[0]))
    return np.stack(rows)


def unroot_like_fasttree(root):
    """A two-child root as FastTree 2.1.11 prints it after ``-intree``: unrooted at a trifurcation.  The root's first child
    is dissolved (its children become the root's, after the second child, which keeps the one length of the root
    edge); where the first child is a leaf, the second is dissolved instead.  Convention read off the binary the
    reference bundles (apples/tools/FastTree-linux) on seven small trees: tests/test_reestimate.py holds them.
    `root.children[0].length` carries the root edge's one estimate on entry (native_lengths)."""
    x, y = root.children
    m_len = x.length
    if x.children:
        keep, gone = y, x
        root.children = [y] + list(x.children)
    else:
        keep, gone = x, y
        root.children = [x] + list(y.children)
    for c in gone.children:
        c.parent = root
    keep.length = m_len
    gone.children = []
    gone.parent = None


def native_lengths(root, ref_fp, protein, device=0):
    """Sets every branch of `root` (binary below the root, two or three children at it) to this build's
    minimum-evolution estimate, rounded to FastTree's five printed decimals.  A two-child root: the one estimate
    goes on the first child, 0 on the second (the caller splits it)."""
    from . import engine
    nodes, parent, children = flatten(root)
    leaves = [v for v in range(len(nodes)) if not children[v]]
    rows = _read_rows(ref_fp, [nodes[v].label for v in leaves])
    leaf_row = [-1] * len(nodes)
    for i, v in enumerate(leaves):
        leaf_row[v] = i
    got = engine.backbone_lengths(parent, children, leaf_row, rows, protein, device)
    for v, nd in enumerate(nodes):
        nd.length = float('%.5f' % got[v]) if parent[v] >= 0 else None
    if len(root.children) == 2:
        root.children[1].length = 0.0


def reestimate_backbone(options):
    """apples/reestimateBackbone.py:22-118.  Rewrites ``options.tree_fp`` to a temporary Newick file with
    re-estimated branch lengths and returns True."""
    assert options.ref_fp
    exe = find_fasttree(getattr(options, 'fasttree_fp', None))
    start = time.time()
    with open(options.tree_fp) as f:
        root = from_newick(f.read())
    rooted = len(root.children) <= 2  # reestimateBackbone.py:35-38
    root = suppress_unifurcations(root)
    if len(root.children) > 3:  # polytomy at the root (:40-46)
        resolve_polytomies(root)
    else:
        for c in root.children:
            resolve_polytomies(c)
    all_len = True
    stack = list(root.children)
    while stack:
        v = stack.pop()
        if v.length is None:
            all_len = False
            break
        stack.extend(v.children)
    restore = rooted and all_len and len(root.children) == 2
    if restore:  # what identifies the input's root edge (:53-65)
        left, right = root.children
        if left.children:
            two, one, len_two, len_one = [c.first_leaf().label for c in left.children], right.first_leaf().label, left.length, right.length
        else:
            two, one, len_two, len_one = [c.first_leaf().label for c in right.children], left.first_leaf().label, right.length, left.length
    tmp = tempfile.mkdtemp(prefix='apples_bb_')
    if exe is None:
        if len(root.children) == 2:
            in_len = [c.length for c in root.children]
        native_lengths(root, options.ref_fp, options.protein_seqs, getattr(options, 'device', 0))
        if len(root.children) == 2 and restore:  # one estimate for the root edge, split as the input had it (:103-110)
            m_len = root.children[0].length
            share = in_len[0] / (in_len[0] + in_len[1]) if in_len[0] + in_len[1] > 0 else 0.5
            root.children[0].length = m_len * share
            root.children[1].length = m_len * (1 - share)
        elif len(root.children) == 2:
            # nothing to restore (a rooted input without all its lengths, or a root polytomy resolved to two children):
            # the reference places on FastTree's own answer, which is unrooted (:91 is skipped)
            unroot_like_fasttree(root)
        text = to_newick(root)
        logging.info('Backbone branch lengths estimated on the GPU (no FastTree executable in use).')
    else:
        resolved_fp = os.path.join(tmp, 'resolved.nwk')
        with open(resolved_fp, 'w') as f:
            f.write(to_newick(root) + '\n')
        log_fp = os.path.join(tmp, 'fasttree.log')
        logging.info('FastTree log file is located here: %s' % log_fp)
        cmd = [exe, '-nosupport', '-nome', '-noml', '-log', log_fp, '-intree', resolved_fp]
        if not options.protein_seqs:
            cmd.append('-nt')
        with open(options.ref_fp) as rf:
            p = subprocess.run(cmd, stdin=rf, stdout=subprocess.PIPE, stderr=sys.stderr)
        if p.returncode != 0:
            raise RuntimeError('FastTree failed with exit code %d (log: %s)' % (p.returncode, log_fp))
        text = p.stdout.decode('utf-8').strip()
        if restore:  # match the rooting of FastTree's output to the input tree (:91-111)
            ft = from_newick(text)
            by_label = {x.label: x for x in ft.leaves()}
            ft = reroot(ft, by_label[one], None)
            m = mrca(ft, [by_label[x] for x in two])
            m_len = m.length
            ft = reroot(ft, m, m_len / 2 if m_len is not None else None)
            if m_len is not None and len_two + len_one > 0:
                for i in range(2):
                    if ft.children[i] is m:
                        ft.children[i].length = m_len * len_two / (len_two + len_one)
                        ft.children[1 - i].length = m_len * len_one / (len_two + len_one)
            text = to_newick(ft)
    out_fp = os.path.join(tmp, 'backbone_reestimated.nwk')
    with open(out_fp, 'w') as f:
        f.write(text.strip() + '\n')
    options.tree_fp = out_fp
    options.reestimate_tmp_dir = tmp  # the caller removes it once the tree is read (cleanup())
    logging.info('[%s] Reestimated branch lengths in %.3f seconds.' % (time.strftime('%H:%M:%S'), time.time() - start))
    return True


def cleanup(options):
    """Remove the temporary directory of :func:`reestimate_backbone` (tree files, FastTree log)."""
    tmp = getattr(options, 'reestimate_tmp_dir', None)
    if tmp and os.path.isdir(tmp):
        shutil.rmtree(tmp, ignore_errors=True)
        options.reestimate_tmp_dir = None
