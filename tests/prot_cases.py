"""Seeded inputs of the g10 fixtures: a protein backbone + alignment + queries for the command line's default
protein route (-p with clusters; apples/Reference.py:84-157 + apples/PoolRepresentativeWorker.py:33-58 with the
21-symbol alphabet).  Shared by tests/golden/make_goldens.py g10 (which runs the REFERENCE on these inputs) and
the tests (which regenerate the same bytes): nothing but seeds and sizes lives here."""
import os

import numpy as np

N_LEAVES, LENGTH, N_QUERIES = 600, 240, 24
CLADE_MIN, CLADE_MAX = 3, 12
# (f, b) of the selection cases; the first is the command line's default
SELECTION_PARAMS = ((0.2, 25), (0.4, 5), (0.1, 60), (10.0, 1))


def prot_case():
    """(dataset, reference rows as FASTA bytes, query rows as FASTA bytes).  The rows carry what fasta2dic rewrites
    (apples/fasta2dic.py:56-67): lower case, the letters BJOUXZ (gaps for -p) and a few non-letters (ordinary symbols,
    which a2i sends to 'A', apples/distance.py:418-678, and the consensus does not count)."""
    from apples_amd import synth
    d = synth.make_dataset(N_LEAVES, LENGTH, N_QUERIES, protein=True, seed_tree=21, seed_aln=22, seed_query=23)
    rng = np.random.default_rng(24)

    def odd(rows):
        rows = rows.copy()
        lower = rng.random(rows.shape) < 0.05
        letters = (rows >= ord('A')) & (rows <= ord('Z'))
        rows[lower & letters] |= 0x20
        for sym, rate in ((b'X', 0.01), (b'B', 0.004), (b'Z', 0.004), (b'*', 0.003), (b'?', 0.002)):
            rows[rng.random(rows.shape) < rate] = sym[0]
        return rows

    ref = odd(d.ref_seqs)
    qry = odd(d.query_seqs)
    qry[5] = ref[17]  # an exact duplicate of a reference row: scoredist gives -0.0, which passes `== 0` (PoolQueryWorker.py:72-75)
    return d, ref, qry


def write_case(dirname):
    """ref.fa, query.fa, backbone.nwk of the case under ``dirname``; returns their paths."""
    d, ref, qry = prot_case()
    paths = [os.path.join(dirname, n) for n in ('prot_ref.fa', 'prot_query.fa', 'prot_backbone.nwk')]
    for path, names, rows in ((paths[0], d.ref_names, ref), (paths[1], d.query_names, qry)):
        with open(path, 'w') as f:
            for n, r in zip(names, rows):
                f.write('>%s\n%s\n' % (n, bytes(r).decode()))
    with open(paths[2], 'w') as f:
        f.write(d.newick + '\n')
    return paths
