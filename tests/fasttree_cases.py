"""Seeded inputs of the g9 FastTree fixtures (shared by tests/golden/make_goldens.py g9 and the tests)."""


def fasttree_cases():
    # (leaves, sites, protein, seed, odd symbols, mean branch length); the *_saturated sets have branches long enough
    # that many profile pairs reach FastTree's LogCorrect cap (raw distance >= 0.74 nt / 0.99 aa, and 3.0 at most)
    return {'nt_rooted': (300, 500, False, 3, True, 0.01), 'aa_rooted': (200, 400, True, 4, False, 0.01),
            'nt_saturated': (80, 300, False, 7, False, 0.7), 'aa_saturated': (80, 300, True, 8, False, 2.5)}


def fasttree_case(n, L, protein, seed, odd, mean_len=0.01):
    """Seeded input of a g9 fixture (the tests call this too)."""
    import numpy as np
    from apples_amd import synth
    d = synth.make_dataset(n, L, 1, protein=protein, seed_tree=seed, gap_rate=0.25 if odd else 0.05, mean_len=mean_len)
    seqs = d.ref_seqs.copy()
    if odd:  # lower case, N (a gap to FastTree), U (= T)
        rng = np.random.default_rng(seed)
        seqs[rng.random(seqs.shape) < 0.02] = ord('N')
        m = rng.random(seqs.shape) < 0.1
        seqs[m] = seqs[m] | 0x20
        seqs[(seqs == ord('T')) & (rng.random(seqs.shape) < 0.3)] = ord('U')
    return d, seqs
