"""APPLES database cache: what the reference pickles in build_applesdtb.py:23-28 --
the indexed tree, its extended Newick string and the reduced reference -- as one versioned numpy
``.npz`` of plain arrays (no pickled classes), so that repeated runs skip the Newick parse, the FASTA
scan, the clustering and the consensus step.  Written by ``build_applesdtb.py``, read by
``run_apples.py -a``."""
import numpy as np

from .fasta import Alignment
from .reference import ReducedReference
from .tree import Tree

FORMAT = 'apples-mi355x-dtb'
VERSION = 1


def save(path, tree, newick, reference, filt_threshold):
    aln = reference.aln
    ca = reference.cluster_arrays()
    labels = tree.labels
    arrays = dict(
        format=np.array(FORMAT), version=np.array(VERSION), protein=np.array(bool(reference.protein)),
        filt_threshold=np.array(float(filt_threshold)),
        parent=tree.parent, edge_len=tree.edge_len, has_len=tree.has_len, child_off=tree.child_off,
        child_idx=tree.child_idx, level=tree.level, is_rooted=np.array(bool(tree.is_rooted)),
        labels=np.array(['' if x is None else x for x in labels]), label_none=np.array([x is None for x in labels]),
        newick=np.array(newick), names=np.array(aln.names), seqs=aln.seqs,
        clustered=np.array(ca is not None))
    if ca is not None:
        arrays.update(cons=ca[0], rep_row=ca[1], member_off=ca[2], member_row=ca[3])
    if reference.eng_rows is not None:  # -s held rows beyond the clustered ones (cluster arrays number the kept rows)
        arrays.update(eng_rows=np.asarray(reference.eng_rows, np.int64))
    with open(path, 'wb') as f:  # (np.savez appends .npz to a path without it: keep the caller's name)
        np.savez(f, **arrays)


def load(path):
    """-> (tree, extended newick string, ReducedReference, protein flag, filter threshold at build time)"""
    z = np.load(path, allow_pickle=False)
    if 'format' not in z.files or str(z['format']) != FORMAT:
        raise ValueError('%s is not an APPLES database of this build (databases pickled by the reference '
                         'implementation cannot be read; rebuild with build_applesdtb.py)' % path)
    if int(z['version']) != VERSION:
        raise ValueError('%s: database version %d, this build reads version %d' % (path, int(z['version']), VERSION))
    none = z['label_none']
    labels = [None if n else str(x) for x, n in zip(z['labels'], none)]
    tree = Tree(z['parent'], z['edge_len'], z['has_len'], labels, z['child_off'], z['child_idx'], z['level'],
                bool(z['is_rooted']))
    aln = Alignment([str(x) for x in z['names']], z['seqs'])
    ref = ReducedReference(aln, bool(z['protein']), None)
    if 'eng_rows' in z.files:
        ref._restrict(z['eng_rows'])
    if bool(z['clustered']):
        ref.cons = z['cons']
        ref.rep_row, ref.member_off, ref.member_row = z['rep_row'], z['member_off'], z['member_row']
    return tree, str(z['newick']), ref, bool(z['protein']), float(z['filt_threshold'])
