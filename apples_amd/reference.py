"""Reduced reference: alignment rows + clusters with consensus representatives.

Mirrors what ReducedReference holds for get_obs_dist (apples/Reference.py:84-115):
``refs`` and ``representatives = [(consensus_or_sequence, [member names]), ...]``.  The
reference obtains clusters from the external TreeCluster tool (Reference.py:87-88), which is not
part of this build: clusters come from a TreeCluster output file (--clusters) or default to
all singletons, for which a representative is the sequence itself
(apples/PoolRepresentativeWorker.py:99-101)."""
import itertools

import numpy as np

NT_ALPHABET = np.frombuffer(b'ACGT-', dtype=np.uint8)
AA_ALPHABET = np.frombuffer(b'ACDEFGHIKLMNPQRSTVWY-', dtype=np.uint8)


def consensus(group_seqs, protein):
    """Per-column most frequent alphabet symbol, ties to the first in alphabet order
    (apples/PoolRepresentativeWorker.py:17-85); symbols outside the alphabet are not counted."""
    alphabet = AA_ALPHABET if protein else NT_ALPHABET
    arr = np.asarray(group_seqs, dtype=np.uint8)
    freq = np.stack([(arr == a).sum(axis=0) for a in alphabet])
    return alphabet[np.argmax(freq, axis=0)]


def _consensus_rows(seqs, mrow, groups, protein):
    """Consensus rows of the member ranges ``groups`` of ``mrow``: libapples_io.so's apples_consensus (threads over clusters:
    0.03 s for 5 000 clusters of a 200 000 x 1 000 alignment where :func:`consensus` cluster by cluster takes 0.4 s), else numpy."""
    L = seqs.shape[1] if seqs.ndim == 2 else 0
    if not groups:
        return np.zeros((0, L), np.uint8)
    from .fasta import _load_io
    lib = _load_io()
    if lib is not None and hasattr(lib, 'apples_consensus') and seqs.flags['C_CONTIGUOUS']:
        import ctypes
        import os
        member = np.concatenate([np.asarray(mrow[a:b], np.int32) for a, b in groups])
        off = np.zeros(len(groups) + 1, np.int64)
        np.cumsum([b - a for a, b in groups], out=off[1:])
        alphabet = AA_ALPHABET if protein else NT_ALPHABET
        out = np.empty((len(groups), L), np.uint8)
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        lib.apples_consensus.restype = ctypes.c_int
        lib.apples_consensus.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32]
        threads = min(32, len(os.sched_getaffinity(0))) if hasattr(os, 'sched_getaffinity') else 0
        if lib.apples_consensus(ptr(seqs), L, ptr(member), ptr(off), len(groups), ptr(alphabet), len(alphabet), ptr(out), threads) == 0:
            return out
    return np.array([consensus(seqs[mrow[a:b]], protein) for a, b in groups], np.uint8).reshape(-1, L)


def read_treecluster(path):
    """[(cluster id, [names])] as the reference groups them: skip the header, sort by cluster id
    as a string (stable), group (apples/Reference.py:93-100)."""
    with open(path) as f:
        f.readline()
        lines = [ln.strip().split('\t') for ln in f.readlines() if ln.strip()]
    lines.sort(key=lambda x: x[1])
    return [(k, [i[0] for i in g]) for k, g in itertools.groupby(lines, lambda x: x[1])]


class ReducedReference:
    def __init__(self, alignment, protein, clusters=None):
        """alignment: apples_amd.fasta.Alignment.  clusters: None or [(id, [names])].

        The reference builds its representatives from TreeCluster's output alone (Reference.py:94-107),
        which names backbone leaves: alignment rows that belong to no cluster are never compared with a
        query, though their names still decide which rows of an extended alignment are queries
        (run_apples.py:85-89).  So ``aln`` keeps every row, and ``eng_aln`` -- what the device holds --
        only the clustered ones (the same object when every row is clustered)."""
        self.aln = alignment
        self.eng_aln = alignment
        self.eng_rows = None  # rows of ``aln`` in ``eng_aln`` (None = all of them, in order)
        self.protein = protein
        self.cons = np.zeros((0, alignment.length), np.uint8)
        self.rep_row = None
        self.member_off = None
        self.member_row = None
        if clusters is not None:
            full_rows = [alignment.index[n] for _, group in clusters for n in group]  # KeyError as Reference.py:149
            if len(set(full_rows)) != len(full_rows):
                raise ValueError('a reference sequence is listed in more than one cluster')
            if len(full_rows) != len(alignment):
                self._restrict(np.array(sorted(full_rows), np.int64))
            index = self.eng_aln.index
            seqs = self.eng_aln.seqs
            rep_row, moff, mrow = [], [0], []
            groups = []  # member ranges of the clusters that get a consensus row, in representative order
            for key, group in clusters:
                rows = [index[n] for n in group]
                if key == '-1':  # singletons pass through (PoolRepresentativeWorker.py:99-101)
                    for r in rows:
                        rep_row.append(r)
                        mrow.append(r)
                        moff.append(len(mrow))
                else:
                    rep_row.append(len(self.eng_aln) + len(groups))
                    groups.append((len(mrow), len(mrow) + len(rows)))
                    mrow += rows
                    moff.append(len(mrow))
            self.cons = _consensus_rows(seqs, mrow, groups, protein)
            self.rep_row = np.array(rep_row, np.int32)
            self.member_off = np.array(moff, np.int32)
            self.member_row = np.array(mrow, np.int32)

    def _restrict(self, rows):
        from .fasta import Alignment
        self.eng_rows = rows
        self.eng_aln = Alignment([self.aln.names[i] for i in rows], self.aln.seqs[rows])

    def cluster_arrays(self):
        """(consensus rows, rep_row, member_off, member_row) in ``eng_aln`` row numbers, or None."""
        if self.rep_row is None:
            return None
        return self.cons, self.rep_row, self.member_off, self.member_row
