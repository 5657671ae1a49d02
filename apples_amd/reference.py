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
        """alignment: apples_amd.fasta.Alignment.  clusters: None or [(id, [names])]."""
        self.aln = alignment
        self.protein = protein
        self.cons = np.zeros((0, alignment.length), np.uint8)
        self.rep_row = None
        self.member_off = None
        self.member_row = None
        if clusters is not None:
            cons, rep_row, moff, mrow = [], [], [0], []
            for key, group in clusters:
                rows = [alignment.index[n] for n in group]
                if key == '-1':  # singletons pass through (PoolRepresentativeWorker.py:99-101)
                    for r in rows:
                        rep_row.append(r)
                        mrow.append(r)
                        moff.append(len(mrow))
                else:
                    rep_row.append(len(alignment) + len(cons))
                    cons.append(consensus(alignment.seqs[rows], protein))
                    mrow += rows
                    moff.append(len(mrow))
            if len(mrow) != len(alignment) or len(set(mrow)) != len(mrow):
                raise ValueError('clusters must cover every reference sequence exactly once')
            self.cons = np.array(cons, np.uint8).reshape(-1, alignment.length)
            self.rep_row = np.array(rep_row, np.int32)
            self.member_off = np.array(moff, np.int32)
            self.member_row = np.array(mrow, np.int32)

    def cluster_arrays(self):
        if self.rep_row is None:
            return None
        return self.cons, self.rep_row, self.member_off, self.member_row
