"""CPU restatement of the APPLES per-query hot path.  TEST INFRASTRUCTURE ONLY.

This module is the parity oracle: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``apples_amd``) never does -- it fails loudly when the HIP library is missing.

Every function restates, in this build's own words and on this build's array
tree (``apples_amd.tree.Tree``), the algorithm of the reference file:line it
cites (paths relative to /root/reference).  It is deliberately written the way
the reference runs -- numpy for the byte arithmetic, plain Python floats for the
tree sums, one query at a time, a fork pool over queries -- so that timing it
is timing the reference's CPU path.

Pinning: ``tests/golden/make_goldens.py`` imports the reference itself in the
build container and dumps fixtures G1-G6; ``tests/test_oracle_golden.py``
checks this module against every one of them.
"""
import heapq
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DATA = os.path.join(os.path.dirname(_HERE), 'apples_amd', 'data', 'blosum45_dist.txt')

# apples/distance.py:12-415 (table) and :418-678 (ASCII -> index, everything else -> 0)
BLOSUM45 = np.loadtxt(_DATA).reshape(400)
AA_ORDER = 'ARNDCQEGHILKMFPSTWYV'
A2I = np.zeros(256, dtype=np.int64)
for _i, _c in enumerate(AA_ORDER):
    A2I[ord(_c)] = _i
    A2I[ord(_c.lower())] = _i
DASH = ord('-')


# --------------------------------------------------------------------------- distances
def pair_counts(a, b):
    """(mismatches, valid) over sites where neither is '-' (apples/distance.py:733-737)."""
    nondash = np.logical_and(a != DASH, b != DASH)
    valid = int(np.count_nonzero(nondash))
    mism = int(np.count_nonzero(np.logical_and(a != b, nondash)))
    return mism, valid


def jc69_from_counts(mism, valid, length, overlap_frac):
    """apples/distance.py:734-745 given the two integer counts."""
    if not valid or valid / length < overlap_frac:
        return -1.0
    p = mism * 1.0 / valid
    if p - np.finfo(float).eps < 0:
        return 0.0
    loc = 1 - (4 * p / 3)
    if 0 >= loc:
        return -1.0
    return -0.75 * np.log(loc)


def jc69(a, b, overlap_frac):
    """apples/distance.py:718-745."""
    mism, valid = pair_counts(a, b)
    return jc69_from_counts(mism, valid, len(a), overlap_frac)


def scoredist(a, b, overlap_frac):
    """apples/distance.py:681-715 (sum via numpy dot of bool x fp64, as the reference)."""
    nondash = np.logical_and(a != DASH, b != DASH)
    valid = np.count_nonzero(nondash)
    if not valid or valid / len(nondash) < overlap_frac:
        return -1.0
    idx = 20 * A2I[a] + A2I[b]
    tot = np.sum(np.dot(nondash, BLOSUM45[idx]))
    if 0 >= 1 - tot / valid:
        return -1.0
    cd = -np.log(1 - tot / valid)
    return cd * 1.3


def scoredist_sequential(a, b, overlap_frac):
    """Same as :func:`scoredist` but summing sites left to right in fp64 -- the
    order the HIP kernel and the C oracle use (BLAS order is CPU specific)."""
    nondash = np.logical_and(a != DASH, b != DASH)
    valid = int(np.count_nonzero(nondash))
    if not valid or valid / len(nondash) < overlap_frac:
        return -1.0
    vals = BLOSUM45[20 * A2I[a] + A2I[b]] * nondash
    tot = 0.0
    for v in vals.tolist():
        tot += v
    if 0 >= 1 - tot / valid:
        return -1.0
    return float(-np.log(1 - tot / valid) * 1.3)


# --------------------------------------------------------------------------- observed set
def get_obs_dist(query, representatives, ref_rows, dist_function, threshold, baseobs, overlap_frac):
    """apples/Reference.py:117-157.

    ``representatives`` = list of (sequence, [member keys]); ``ref_rows`` maps a
    member key to its sequence.  Returns the insertion-ordered dict.
    """
    obs = {}
    obs_num = 0
    heap = []
    for i, (cons, _group) in enumerate(representatives):
        d = dist_function(query, cons, overlap_frac)
        if d >= 0:
            heap.append((d, i))
    heapq.heapify(heap)
    while heap:
        d, i = heapq.heappop(heap)
        if d <= threshold or obs_num < baseobs:
            for key in representatives[i][1]:
                dm = dist_function(query, ref_rows[key], overlap_frac)
                if not dm < 0:
                    obs[key] = dm
                    obs_num += 1
        else:
            break
    return obs


def valid_dists(obs_dist, in_tree, baseobs, threshold):
    """The ``-d`` filter, apples/PoolQueryWorker.py:44-59."""
    out = {}
    tx = 0
    for k, v in sorted(obs_dist.items(), key=lambda kv: kv[1]):
        if v < 0 or k not in in_tree:
            continue
        tx += 1
        if tx > baseobs and v > threshold:
            break
        out[k] = v
    return out


# --------------------------------------------------------------------------- induced subtree
def induced_subtree(tree, obs_nodes):
    """apples/Subtree.py:23-43 + apples/PrioritySet.py.  ``obs_nodes`` = node ids
    of the observed leaves.  Returns (valid bool array, lca, num_nodes)."""
    valid = np.zeros(tree.n_nodes, dtype=bool)
    heap = []
    inset = set()
    for v in obs_nodes:
        if v not in inset:
            heapq.heappush(heap, (-int(tree.level[v]), int(v)))
            inset.add(v)
    count = 0
    while len(heap) > 1:
        _, x = heapq.heappop(heap)
        inset.remove(x)
        valid[x] = True
        count += 1
        p = int(tree.parent[x])
        if p not in inset:
            heapq.heappush(heap, (-int(tree.level[p]), p))
            inset.add(p)
    return valid, heap[0][1], count


def _valid_postorder(valid):
    # valid nodes in post-order == ascending edge_index (apples/Subtree.py:56-70, util.py:65-69)
    return np.nonzero(valid)[0].tolist()


# --------------------------------------------------------------------------- S / R sweeps
def _lift_ols(t, e):
    # what a parent adds for a child/sibling tuple over its edge e (apples/OLS.py:36-44)
    S, Sd, Sd2, SDd, SD2, SD = t
    return (S, S * e + Sd, S * e * e + Sd2 + 2 * e * Sd, e * SD + SDd, SD2, SD)


def _lift_fm(t, e):
    # apples/FM.py:31-40 ; tuple (S, Sd_D, Sd_D2, Sd2_D2, S1_D, S1_D2)
    S, Sd_D, Sd_D2, Sd2_D2, S1_D, S1_D2 = t
    return (S, e * S1_D + Sd_D, e * S1_D2 + Sd_D2, S1_D2 * e * e + Sd2_D2 + 2 * e * Sd_D2, S1_D, S1_D2)


def _lift_be(t, e):
    # apples/BE.py:20-30 ; tuple (S, Sd, Sd_D, Sd2_D, SD, S1_D)
    S, Sd, Sd_D, Sd2_D, SD, S1_D = t
    return (S, S * e + Sd, e * S1_D + Sd_D, S1_D * e * e + Sd2_D + 2 * e * Sd_D, SD, S1_D)


def _leaf_tuple(method, D):
    if method == 'OLS' or method == 'BME':  # apples/OLS.py:27-33, BME.py:11-17
        return (1, 0, 0, 0, D * D, D)
    if method == 'FM':  # apples/FM.py:21-27
        return (1, 0, 0, 0, 1.0 / D, 1.0 / (D * D))
    return (1, 0, 0, 0, D, 1.0 / D)  # BE, apples/BE.py:11-17


_LIFT = {'OLS': _lift_ols, 'BME': _lift_ols, 'FM': _lift_fm, 'BE': _lift_be}


def s_values(tree, valid, leaf_dist, method):
    """all_S_values: apples/OLS.py:12-44, FM.py:6-40, BE.py:6-30, BME.py:6-30."""
    lift = _LIFT[method]
    S = {}
    for v in _valid_postorder(valid):
        ch = tree.children(v)
        if len(ch) == 0:
            S[v] = _leaf_tuple(method, leaf_dist[v])
        else:
            acc = [0, 0, 0, 0, 0, 0]
            vch = [int(c) for c in ch if valid[c]]
            coef = 1 / len(vch) if method == 'BME' else None
            for c in vch:
                t = lift(S[c], float(tree.edge_len[c]))
                for k in range(6):
                    acc[k] += t[k] if coef is None else coef * t[k]
            S[v] = tuple(acc)
    return S


def r_values(tree, valid, lca, S, method):
    """all_R_values: apples/OLS.py:46-80, FM.py:42-76, BE.py:32-57, BME.py:32-60."""
    lift = _LIFT[method]
    R = {}
    for v in reversed(_valid_postorder(valid)):  # parents before children
        p = int(tree.parent[v])
        acc = [0, 0, 0, 0, 0, 0]
        sibs = [int(c) for c in tree.children(p) if valid[c] and c != v]
        coef = None
        if method == 'BME':
            nonroot = 1 if p != lca else 0
            coef = 1 / (nonroot + len(sibs))
        for s in sibs:
            t = lift(S[s], float(tree.edge_len[s]))
            for k in range(6):
                acc[k] += t[k] if coef is None else coef * t[k]
        if p != lca and valid[p]:
            t = lift(R[p], float(tree.edge_len[p]))
            for k in range(6):
                acc[k] += t[k] if coef is None else coef * t[k]
        R[v] = tuple(acc)
    return R


# --------------------------------------------------------------------------- 2x2 solve, residual
def solve2_2(e, a11, a12, a21, a22, c1, c2, negative_branch):
    """apples/util.py:6-54.  Returns (x_1, x_2, x_1_neg, x_2_neg); clamped values stay int 0."""
    det = 1 / (a11 * a22 - a12 * a21)
    assert det != 0
    x1n = (a22 * c1 - a12 * c2) * det
    x2n = (-a21 * c1 + a11 * c2) * det
    x1, x2 = x1n, x2n
    if not negative_branch:
        if x1n < 0 and x2n < 0:
            x1 = 0
            x2 = 0
        elif x1n > 0 and x2n < 0:
            x1 = max(c1 * 1.0 / a11, 0)
            x2 = 0
        elif x1n < 0 and 0 <= x2n and x2n <= e:
            x1 = 0
            x2 = min(max(c2 * 1.0 / a22, 0), e)
        elif x1n < 0 and x2n > e:
            x1 = 0
            x2 = e
        elif x1n > 0 and x2n > e:
            x1 = max((c1 * 1.0 - a12 * e) / a11, 0)
            x2 = e
    return x1, x2, x1n, x2n


def _system(method, s, r, e):
    # placement_per_edge: apples/OLS.py:90-96, FM.py:86-92, BE.py:61-67, BME.py:64-70
    if method == 'OLS' or method == 'BME':
        S, Sd, Sd2, SDd, SD2, SD = s
        R, Rd, Rd2, RDd, RD2, RD = r
        a11 = R + S
        a12 = R - S
        c1 = RD + SD - e * S - Rd - Sd
        c2 = RD - SD + e * S - Rd + Sd
    elif method == 'FM':
        S, Sd_D, Sd_D2, Sd2_D2, S1_D, S1_D2 = s
        R, Rd_D, Rd_D2, Rd2_D2, R1_D, R1_D2 = r
        a11 = R1_D2 + S1_D2
        a12 = R1_D2 - S1_D2
        c1 = R1_D + S1_D - e * S1_D2 - Rd_D2 - Sd_D2
        c2 = R1_D - S1_D + e * S1_D2 - Rd_D2 + Sd_D2
    else:  # BE
        S, Sd, Sd_D, Sd2_D, SD, S1_D = s
        R, Rd, Rd_D, Rd2_D, RD, R1_D = r
        a11 = R1_D + S1_D
        a12 = R1_D - S1_D
        c1 = R + S - e * S1_D - Rd_D - Sd_D
        c2 = R - S + e * S1_D - Rd_D + Sd_D
    return a11, a12, a12, a11, c1, c2


def error_per_edge(method, s, r, e, x1, x2):
    """apples/OLS.py:100-128, FM.py:96-124, BE.py:71-80, BME.py:74-83 (``**2`` is libm pow there)."""
    if method == 'OLS' or method == 'BME':
        S, Sd, Sd2, SDd, SD2, SD = s
        R, Rd, Rd2, RDd, RD2, RD = r
        A = RD2 + SD2
        B = 2 * (x1 + x2) * Rd + 2 * (e + x1 - x2) * Sd
        C = (x1 + x2) ** 2 * R + (e + x1 - x2) ** 2 * S
        D = -2 * (x1 + x2) * RD - 2 * (e + x1 - x2) * SD
        E = -2 * RDd - 2 * SDd
        F = Rd2 + Sd2
    elif method == 'FM':
        S, Sd_D, Sd_D2, Sd2_D2, S1_D, S1_D2 = s
        R, Rd_D, Rd_D2, Rd2_D2, R1_D, R1_D2 = r
        A = R + S
        B = 2 * (x1 + x2) * Rd_D2 + 2 * (e + x1 - x2) * Sd_D2
        C = (x1 + x2) ** 2 * R1_D2 + (e + x1 - x2) ** 2 * S1_D2
        D = -2 * (x1 + x2) * R1_D - 2 * (e + x1 - x2) * S1_D
        E = -2 * Rd_D - 2 * Sd_D
        F = Rd2_D2 + Sd2_D2
    else:  # BE
        S, Sd, Sd_D, Sd2_D, SD, S1_D = s
        R, Rd, Rd_D, Rd2_D, RD, R1_D = r
        A = RD + SD
        B = 2 * (x1 + x2) * Rd_D + 2 * (e + x1 - x2) * Sd_D
        C = (x1 + x2) ** 2 * R1_D + (e + x1 - x2) ** 2 * S1_D
        D = -2 * (x1 + x2) * R - 2 * (e + x1 - x2) * S
        E = -2 * Rd - 2 * Sd
        F = Rd2_D + Sd2_D
    return A + B + C + D + E + F


def per_edge(tree, valid, S, R, method, negative_branch):
    """placement_per_edge + error for every valid node -> {v: (x1, x2, x1n, x2n, err)}."""
    out = {}
    for v in _valid_postorder(valid):
        e = float(tree.edge_len[v])
        a11, a12, a21, a22, c1, c2 = _system(method, S[v], R[v], e)
        x1, x2, x1n, x2n = solve2_2(e, a11, a12, a21, a22, c1, c2, negative_branch)
        out[v] = (x1, x2, x1n, x2n, error_per_edge(method, S[v], R[v], e, x1, x2))
    return out


def placement(tree, valid, num_nodes, edges, criterion):
    """apples/Algorithm.py:62-101.  ``edges`` from :func:`per_edge`."""
    order = _valid_postorder(valid)
    if criterion == 'HYBRID':
        sm = heapq.nsmallest(math.floor(math.log2(num_nodes)), order, key=lambda v: edges[v][4])
        best = min(sm, key=lambda v: edges[v][0])
    elif criterion == 'ME':
        best = min(order, key=lambda v: edges[v][0])
    else:
        best = min(order, key=lambda v: edges[v][4])
    x1, x2, _, _, err = edges[best]
    e = float(tree.edge_len[best])
    flag = 1 if (x1 == 0 and err > 0 and (x2 == 0 or x2 == e)) else 0
    return [int(best), err, 1, e - x2, x1], flag


# --------------------------------------------------------------------------- per-query driver
def place_observed(tree, obs, method, criterion, negative_branch):
    """apples/PoolQueryWorker.py:101-133 on an observed ``{leaf name: distance}`` dict."""
    nodes = [tree.name_to_node[k] for k in obs if k in tree.name_to_node]
    leaf_dist = {tree.name_to_node[k]: v for k, v in obs.items() if k in tree.name_to_node}
    valid, lca, num_nodes = induced_subtree(tree, nodes)
    if method not in ('BE', 'FM', 'BME'):
        method = 'OLS'
    S = s_values(tree, valid, leaf_dist, method)
    R = r_values(tree, valid, lca, S, method)
    edges = per_edge(tree, valid, S, R, method, negative_branch)
    return placement(tree, valid, num_nodes, edges, criterion)


def runquery(tree, query_name, obs, method='FM', criterion='MLSE', negative_branch=False,
             exclude_intplace=False):
    """apples/PoolQueryWorker.py:28-141 after the observed dict exists.  Returns the
    per-query jplace dict."""
    jplace = {'placements': [{'p': [[0, 0, 1, 0, 0]], 'n': [query_name]}]}
    obs = dict(obs)
    if query_name in tree.name_to_node:  # :63-70
        if query_name in obs:
            del obs[query_name]
        query_name = query_name + '-query'
        jplace['placements'][0]['n'] = [query_name]
    for k, v in obs.items():  # :72-75
        if v == 0:
            jplace['placements'][0]['p'][0][0] = tree.name_to_node[k]
            return jplace
    if len(obs) <= 2:  # :97-98
        jplace['placements'][0]['p'][0][0] = -1
        return jplace
    presult, flag = place_observed(tree, obs, method, criterion, negative_branch)
    jplace['placements'][0]['p'] = [presult]
    if flag == 1 and exclude_intplace:  # :120-125
        jplace['placements'][0]['p'][0][0] = -1
    return jplace


def join_jplace(lst):
    """apples/jutil.py:1-19 (keeps the first result even when unplaceable)."""
    result = lst[0]
    if len(lst) == 1:
        if result['placements'][0]['p'][0][0] == -1:
            result['placements'] = []
    else:
        for i in range(1, len(lst)):
            if lst[i]['placements'][0]['p'][0][0] != -1:
                result['placements'] = result['placements'] + lst[i]['placements']
    return result


# --------------------------------------------------------------------------- pool driver (CPU baseline)
class _Worker:
    tree = None
    reps = None
    rows = None
    dist_fn = None
    params = None

    @classmethod
    def run(cls, name, seq):
        p = cls.params
        obs = get_obs_dist(seq, cls.reps, cls.rows, cls.dist_fn, p['threshold'], p['baseobs'], p['overlap'])
        return runquery(cls.tree, name, obs, p['method'], p['criterion'], p['negative'], p['exclude'])


def run_pool(tree, ref_names, ref_seqs, query_names, query_seqs, protein=False, method='FM', criterion='MLSE',
             threshold=0.2, baseobs=25, overlap=0.001, negative=False, exclude=False, clusters=None, threads=1):
    """The reference's driver shape (run_apples.py:93-102): fork pool, starmap over queries.
    ``clusters`` = list of (consensus row, [member names]); None = all singletons
    (apples/PoolRepresentativeWorker.py:99-101)."""
    import multiprocessing as mp
    rows = {n: ref_seqs[i] for i, n in enumerate(ref_names)}
    reps = clusters if clusters is not None else [(ref_seqs[i], [n]) for i, n in enumerate(ref_names)]
    _Worker.tree = tree
    _Worker.reps = reps
    _Worker.rows = rows
    _Worker.dist_fn = scoredist if protein else jc69
    _Worker.params = dict(threshold=threshold, baseobs=baseobs, overlap=overlap, method=method,
                          criterion=criterion, negative=negative, exclude=exclude)
    tasks = [(n, query_seqs[i]) for i, n in enumerate(query_names)]
    if threads <= 1:
        return [_Worker.run(*t) for t in tasks]
    ctx = mp.get_context('fork')
    with ctx.Pool(threads) as pool:
        return pool.starmap(_Worker.run, tasks)


def _noop(_):
    return 0


class _RowReps:
    """All-singleton representatives over one 2-D byte matrix: what ``get_obs_dist`` iterates and indexes
    (``(sequence, [member keys])``), with the row number as the member key.  The timed pool uses it
    instead of 200 k (array, [name]) tuples: forked workers would copy-on-write every page holding
    those objects the first time they walk the list (reference counts are written on every access),
    which at 200 k references costs each worker more than the queries themselves."""

    def __init__(self, seqs):
        self.seqs = seqs

    def __len__(self):
        return len(self.seqs)

    def __getitem__(self, i):
        return self.seqs[i], (i,)

    def __iter__(self):
        seqs = self.seqs
        for i in range(len(seqs)):
            yield seqs[i], (i,)


class _RowWorker:
    tree = None
    seqs = None
    names = None
    dist_fn = None
    params = None

    @classmethod
    def run(cls, name, seq):
        p = cls.params
        obs = get_obs_dist(seq, _RowReps(cls.seqs), cls.seqs, cls.dist_fn, p['threshold'], p['baseobs'], p['overlap'])
        names = cls.names
        obs = {names[i]: d for i, d in obs.items()}  # same insertion order as with name keys
        return runquery(cls.tree, name, obs, p['method'], p['criterion'], p['negative'], p['exclude'])


def time_pool(tree, ref_names, ref_seqs, query_names, query_seqs, threads, **kw):
    """Steady-state timing of the pool driver for bench.py's cpu_baseline: the fork pool is started
    and warmed first (its start-up, which the reference's own "Processed all queries" timer includes,
    is returned separately), then the starmap over the sample is timed (run_apples.py:101-102; one task
    per chunk so that no worker sits on a queue of slow queries).
    Returns (seconds_steady, seconds_startup, results)."""
    import multiprocessing as mp
    import time
    _RowWorker.tree = tree
    _RowWorker.seqs = ref_seqs
    _RowWorker.names = list(ref_names)
    _RowWorker.dist_fn = scoredist if kw.get('protein') else jc69
    _RowWorker.params = dict(threshold=kw.get('threshold', 0.2), baseobs=kw.get('baseobs', 25),
                             overlap=kw.get('overlap', 0.001), method=kw.get('method', 'FM'),
                             criterion=kw.get('criterion', 'MLSE'), negative=False, exclude=False)
    tasks = [(n, query_seqs[i]) for i, n in enumerate(query_names)]
    if threads <= 1:
        t1 = time.time()
        res = [_RowWorker.run(*t) for t in tasks]
        return time.time() - t1, 0.0, res
    ctx = mp.get_context('fork')
    t0 = time.time()
    with ctx.Pool(threads) as pool:
        pool.map(_noop, range(4 * threads))
        t1 = time.time()
        res = pool.starmap(_RowWorker.run, tasks, chunksize=1)
        t2 = time.time()
    return t2 - t1, t1 - t0, res


class _TableWorker:
    tree = None
    cols = None
    params = None

    @classmethod
    def run(cls, name, row):
        p = cls.params
        obs = valid_dists(dict(zip(cls.cols, row.tolist())), cls.tree.name_to_node, p['baseobs'], p['threshold'])
        return runquery(cls.tree, name, obs, p['method'], p['criterion'], False, False)


def time_pool_table(tree, col_names, query_names, D, threads, method='BME', criterion='MLSE', threshold=0.2, baseobs=25):
    """Distance-table analogue of :func:`time_pool` (run_apples.py -d path)."""
    import multiprocessing as mp
    import time
    _TableWorker.tree = tree
    _TableWorker.cols = list(col_names)
    _TableWorker.params = dict(method=method, criterion=criterion, threshold=threshold, baseobs=baseobs)
    tasks = [(n, D[i]) for i, n in enumerate(query_names)]
    ctx = mp.get_context('fork')
    t0 = time.time()
    with ctx.Pool(threads) as pool:
        pool.map(_noop, range(4 * threads))
        t1 = time.time()
        res = pool.starmap(_TableWorker.run, tasks, chunksize=1)
        t2 = time.time()
    return t2 - t1, t1 - t0, res
