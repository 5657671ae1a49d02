"""Seeded inputs of the g9 FastTree fixtures (shared by tests/golden/make_goldens.py g9 and the tests)."""


def fasttree_cases():
    return {'nt_rooted': (300, 500, False, 3, True), 'aa_rooted': (200, 400, True, 4, False)}


def fasttree_case(n, L, protein, seed, odd):
    """Seeded input of a g9 fixture (the tests call this too)."""
    import numpy as np
    from apples_amd import synth
    d = synth.make_dataset(n, L, 1, protein=protein, seed_tree=seed, gap_rate=0.25 if odd else 0.05)
    seqs = d.ref_seqs.copy()
    if odd:  # lower case, N (a gap to FastTree), U (= T)
        rng = np.random.default_rng(seed)
        seqs[rng.random(seqs.shape) < 0.02] = ord('N')
        m = rng.random(seqs.shape) < 0.1
        seqs[m] = seqs[m] | 0x20
        seqs[(seqs == ord('T')) & (rng.random(seqs.shape) < 0.3)] = ord('U')
    return d, seqs
