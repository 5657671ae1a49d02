#!/usr/bin/env python3
"""Would a lower-bounding table of rank < 20 pay for the scoredist filter (dist_sd.hip:k_sd_gemm)?  CPU probe (numpy), config 4's
synthetic inputs.  A non-negative rank-r lower bound of the zero-diagonal BLOSUM45 dissimilarity table is a sum of bicliques
between disjoint residue groups; the natural family is a reduced alphabet: the reference side group-hot (r groups), the query
side min over the group of its table row.  Counted per form: the pairs the filter would pass on (candidates for the exact
evaluation) and the mean bound / true sum, with the table values exact and rounded down to the fp4 grid the filter multiplies.
Usage: python scripts/r05_sd_rank_probe.py [queries]   (output of the round-5 run: profiles/r05_sd_rank_probe.txt)"""
import sys, time, itertools
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from apples_amd import synth
T = np.array([float(x) for l in open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'apples_amd', 'data', 'blosum45_dist.txt')) if not l.startswith('#') for x in l.split()]).reshape(20,20)
order = b'ARNDCQEGHILKMFPSTWYV'
a2i = np.full(256, 0, np.int64)
for i,c in enumerate(order): a2i[c]=i; a2i[c+32]=i
grid4 = np.array([0,.5,1,1.5,2,3,4,6])/4.0
def down(x, grid):
    return grid[np.searchsorted(grid, x+1e-12, side='right')-1]
nq = int(sys.argv[1]) if len(sys.argv)>1 else 48
d = synth.make_dataset(50000, 500, nq, protein=True)
R = d.ref_seqs; Q = d.query_seqs
rgap = R == ord('-'); qgap = Q == ord('-')
ri = a2i[R]; qi = a2i[Q]
c = 1 - np.exp(-0.2/1.3)
# grouping search: greedy agglomerative on columns
def group_tables(groups):
    # U[a][k] = min over b in group k of T[a][b]
    return np.stack([T[:, g].min(axis=1) for g in groups], axis=1), np.array([next(k for k,g in enumerate(groups) if b in g) for b in range(20)])
def greedy(r, w=None):
    groups = [[b] for b in range(20)]
    while len(groups) > r:
        best=None
        for i,j in itertools.combinations(range(len(groups)),2):
            g = groups[i]+groups[j]
            # loss: sum over a of sum_b in g (T[a][b] - min_b' T[a][b'])
            m = T[:, g].min(axis=1)
            loss = (T[:, g] - m[:,None]).sum()
            if best is None or loss < best[0]: best=(loss,i,j)
        _,i,j = best
        groups[i] = groups[i]+groups[j]; del groups[j]
    return groups
res = {}
t0=time.time()
tabs = {'fp4_onehot': (down(T, grid4), np.arange(20))}
for r in (16, 14, 12, 10, 8):
    g = greedy(r)
    U, col = group_tables(g)
    tabs['grp%d_fp4' % r] = (down(U, grid4), col)
    tabs['grp%d_exact' % r] = (U, col)
    print(r, [''.join(chr(order[b]) for b in gg) for gg in g])
stats = {k: [] for k in tabs}; exact_pass = 0; total = 0; cand = {k:0 for k in tabs}
for q in range(nq):
    both = ~(rgap | qgap[q][None,:])          # [N, L]
    valid = both.sum(axis=1)
    tot = np.where(both, T[qi[q][None,:], ri], 0.0).sum(axis=1)
    nvr = (~rgap).sum(axis=1); nvq = (~qgap[q]).sum()
    cut = c * np.minimum(nvr, nvq)
    ok = valid > 0
    total += ok.sum()
    exact_pass += ((tot <= c*valid) & ok).sum()
    for k,(U,col) in tabs.items():
        lb = np.where(both, U[qi[q][None,:], col[ri]], 0.0).sum(axis=1)
        cand[k] += ((lb <= cut*(1+1e-6)) & ok).sum()
        stats[k].append((lb[ok]/np.maximum(tot[ok],1e-9)).mean())
print('pairs', total, 'pass exact %.4f%%' % (100*exact_pass/total), 'time', time.time()-t0)
for k in tabs:
    print('%-14s candidates %.3f%%  per query %.0f  mean lb/tot %.3f' % (k, 100*cand[k]/total, cand[k]/nq, np.mean(stats[k])))
