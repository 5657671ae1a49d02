#!/usr/bin/env python3
"""Generate the golden fixtures by running the REFERENCE itself (build container only).

Imports /root/reference's own modules (apples.distance, Reference, PoolQueryWorker,
OLS/FM/BME/BE, jutil, util) and records their outputs on fixed inputs.  Refuses to run
when the reference is absent (e.g. on the GPU box): the fixtures it wrote are
committed under tests/golden/ and are all the tests need.

The reference needs a module called ``treeswift`` (not installed here); a minimal
stand-in built on this repo's own Newick reader is registered under that name.  It
contributes no arithmetic -- only node objects with label / edge_length / children /
parent / traverse_postorder.

Usage:  python tests/golden/make_goldens.py
"""
import hashlib
import json
import os
import sys
import types

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DATA = os.path.join(HERE, 'data')

if not os.path.isdir(os.path.join(REF, 'apples')):
    sys.exit('reference not present at %s: goldens can only be regenerated in the build container' % REF)

sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np  # noqa: E402

from apples_amd.tree import parse_newick  # noqa: E402
from apples_amd import synth  # noqa: E402


# ----------------------------------------------------------------------------- treeswift stand-in
class Node:
    def __init__(self, idx):
        self._i = idx
        self.label = None
        self.edge_length = None
        self.children = []
        self.parent = None

    def is_leaf(self):
        return len(self.children) == 0

    def __lt__(self, other):
        return self._i < other._i

    def traverse_postorder(self, leaves=True, internal=True):
        s1, s2 = [self], []
        while s1:
            n = s1.pop()
            s2.append(n)
            s1.extend(n.children)
        while s2:
            n = s2.pop()
            if (leaves and n.is_leaf()) or (internal and not n.is_leaf()):
                yield n


class TsTree:
    def __init__(self, root, is_rooted):
        self.root = root
        self.is_rooted = is_rooted

    def traverse_postorder(self, leaves=True, internal=True):
        return self.root.traverse_postorder(leaves, internal)


def ts_from_text(text):
    t = parse_newick(text)
    nodes = [Node(i) for i in range(t.n_nodes)]
    for i, nd in enumerate(nodes):
        nd.label = t.labels[i]
        nd.edge_length = float(t.edge_len[i]) if t.has_len[i] else None
        nd.children = [nodes[c] for c in t.children(i)]
        if t.parent[i] >= 0:
            nd.parent = nodes[t.parent[i]]
    return TsTree(nodes[t.root], t.is_rooted)


def read_tree(path, schema='newick'):
    with open(path) as f:
        return ts_from_text(f.read())


shim = types.ModuleType('treeswift')
shim.read_tree = read_tree
shim.Node = Node
shim.Tree = TsTree
sys.modules['treeswift'] = shim

from apples import util as ref_util  # noqa: E402
from apples import jutil as ref_jutil  # noqa: E402
from apples.distance import jc69 as ref_jc69, scoredist as ref_scoredist  # noqa: E402
from apples.fasta2dic import fasta2dic as ref_fasta2dic  # noqa: E402
from apples.Reference import ReducedReference  # noqa: E402
from apples.PoolQueryWorker import PoolQueryWorker  # noqa: E402
from apples.Subtree import Subtree  # noqa: E402
from apples.OLS import OLS  # noqa: E402
from apples.FM import FM  # noqa: E402
from apples.BME import BME  # noqa: E402
from apples.BE import BE  # noqa: E402

ALG = {'OLS': OLS, 'FM': FM, 'BME': BME, 'BE': BE}
TUPLES = {
    'OLS': (['S', 'Sd', 'Sd2', 'SDd', 'SD2', 'SD'], ['R', 'Rd', 'Rd2', 'RDd', 'RD2', 'RD']),
    'BME': (['BS', 'BSd', 'BSd2', 'BSDd', 'BSD2', 'BSD'], ['BR', 'BRd', 'BRd2', 'BRDd', 'BRD2', 'BRD']),
    'FM': (['S', 'Sd_D', 'Sd_D2', 'Sd2_D2', 'S1_D', 'S1_D2'], ['R', 'Rd_D', 'Rd_D2', 'Rd2_D2', 'R1_D', 'R1_D2']),
    'BE': (['S', 'Sd', 'Sd_D', 'Sd2_D', 'SD', 'S1_D'], ['R', 'Rd', 'Rd_D', 'Rd2_D', 'RD', 'R1_D']),
}


def prepare_tree(path_or_text, is_text=False):
    tree = ts_from_text(path_or_text) if is_text else read_tree(path_or_text)
    ref_util.index_edges(tree)
    ref_util.set_levels(tree)
    n2n = {}
    for leaf in tree.traverse_postorder(internal=False):
        n2n[leaf.label] = leaf
    return tree, n2n, ref_jutil.extended_newick(tree)


def make_reference(refs, prot, threshold, baseobs, clusters=None):
    """ReducedReference without the TreeCluster subprocess: set exactly the attributes
    get_obs_dist reads (apples/Reference.py:138-149)."""
    r = ReducedReference.__new__(ReducedReference)
    r.refs = refs
    r.prot_flag = prot
    r.dist_function = ref_scoredist if prot else ref_jc69
    r.threshold = threshold
    if clusters is None:
        r.representatives = [(refs[k], [k]) for k in refs]
    else:
        r.representatives = clusters
    r.baseobs = baseobs
    return r


def options(method='FM', criterion='MLSE', negative=False, f=0.2, b=25, overlap=0.001, exclude=False):
    return types.SimpleNamespace(method_name=method, criterion_name=criterion, negative_branch=negative,
                                 filt_threshold=f, base_observation_threshold=b,
                                 minimum_alignment_overlap=overlap, exclude_intplace=exclude)


def run_queries(reference, opts, n2n, queries):
    PoolQueryWorker.set_class_attributes(reference, opts, n2n)
    return [PoolQueryWorker.runquery(*q) for q in queries]


def jdump(obj, name):
    with open(os.path.join(HERE, name), 'w') as f:
        json.dump(obj, f, indent=1, default=_json_default)
        f.write('\n')


def _json_default(o):
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, (np.integer,)):
        return int(o)
    raise TypeError(type(o))


def placements_of(results):
    """[(name, p-row)] with python scalars; ints stay ints so that 0 vs 0.0 survives JSON."""
    out = []
    for r in results:
        pl = r['placements'][0]
        row = [x if isinstance(x, int) and not isinstance(x, bool) else float(x) for x in pl['p'][0]]
        out.append({'n': pl['n'][0], 'p': row})
    return out


def consensus(group, refs, prot):
    from apples.PoolRepresentativeWorker import PoolRepresentativeWorker
    return PoolRepresentativeWorker._find_representative(group, refs, prot)


# ============================================================================= G1 distances
def g1():
    refs = ref_fasta2dic(os.path.join(DATA, 'ref.fa'), False, False)
    qs = ref_fasta2dic(os.path.join(DATA, 'query.fa'), False, False)
    rn, qn = list(refs), list(qs)
    d = np.array([[ref_jc69(qs[q], refs[r], 0.001) for r in rn] for q in qn])
    np.savez_compressed(os.path.join(HERE, 'g1_jc69_data.npz'), dist=d, ref_names=np.array(rn), query_names=np.array(qn))

    # seeded nt pairs with exotic symbols + edge cases
    rng = np.random.default_rng(11)
    L = 257
    sym = np.frombuffer(b'ACGT-', dtype='S1')
    a = sym[rng.choice(5, size=(48, L), p=[.22, .22, .22, .22, .12])]
    b = a.copy()
    mut = rng.random(b.shape) < rng.random((48, 1)) * 0.9
    b[mut] = sym[rng.integers(0, 5, size=int(mut.sum()))]
    b[3, :] = b'-'                      # all gap
    b[4, 5:] = b'-'                     # overlap 5/257 above -V 0.001
    a[5], b[5] = a[6], a[6]             # identical -> 0.0
    b[7] = sym[(np.searchsorted(sym, a[7]) + 1) % 4]  # all different -> saturated
    a[8, ::7] = b'*'; b[8, ::5] = b'?'  # non-letter symbols are ordinary symbols
    a[9, :] = b'-'
    out = {'a': a.view(np.uint8), 'b': b.view(np.uint8)}
    out['jc69_V0.001'] = np.array([ref_jc69(a[i], b[i], 0.001) for i in range(48)])
    out['jc69_V0.5'] = np.array([ref_jc69(a[i], b[i], 0.5) for i in range(48)])
    np.savez_compressed(os.path.join(HERE, 'g1_jc69_synth.npz'), **out)

    # scoredist: seeded aa pairs, 5 % gaps, lower-case and odd symbols
    rng = np.random.default_rng(12)
    L = 300
    aa = np.frombuffer(b'ARNDCQEGHILKMFPSTWYV', dtype='S1')
    A = aa[rng.integers(0, 20, size=(64, L))]
    B = A.copy()
    mut = rng.random(B.shape) < rng.random((64, 1)) * 0.8
    B[mut] = aa[rng.integers(0, 20, size=int(mut.sum()))]
    A[rng.random(A.shape) < 0.05] = b'-'
    B[rng.random(B.shape) < 0.05] = b'-'
    B[0] = A[0]                          # identical -> -0.0
    B[1, :] = b'-'
    A[2, ::9] = b'x'; B[2, ::4] = b'*'; A[2, 1::9] = b'a'  # a2i: unknown -> 'A'(0); lower case maps like upper
    B[3, 3:] = b'-'
    out = {'a': A.view(np.uint8), 'b': B.view(np.uint8)}
    out['scoredist_V0.001'] = np.array([ref_scoredist(A[i], B[i], 0.001) for i in range(64)])
    out['scoredist_V0.5'] = np.array([ref_scoredist(A[i], B[i], 0.5) for i in range(64)])
    np.savez_compressed(os.path.join(HERE, 'g1_scoredist_synth.npz'), **out)

    # fasta2dic encoding rules on a tiny file
    fa = os.path.join(HERE, 'g1_encoding.fa')
    with open(fa, 'w') as f:
        f.write('>s1 some description\nACGTacgtNnUuRYKM-.?*\nBDEFHIJKLMOPQSVWXZ\n>s2\nacgtn-\n>s3 last line has no newline\nACGT')
    enc = {}
    for prot in (False, True):
        for mask in (False, True):
            dct = ref_fasta2dic(fa, prot, mask)
            enc['prot%d_mask%d' % (prot, mask)] = {k: v.tobytes().decode() for k, v in dct.items()}
    jdump(enc, 'g1_encoding.json')


# ============================================================================= G2 selection
def clade_clusters(tree, refs, prot, min_size, max_size):
    """Hand-made multi-member clusters: maximal clades with size in [min,max]; the rest singletons
    ('-1' group first, as sorting cluster ids as strings puts it, apples/Reference.py:97)."""
    size = {}
    for n in tree.traverse_postorder():
        size[n] = 1 if n.is_leaf() else sum(size[c] for c in n.children)
    clusters, single = [], []
    stack = [tree.root]
    while stack:
        n = stack.pop()
        if n.is_leaf():
            single.append(n.label)
        elif min_size <= size[n] <= max_size:
            clusters.append([l.label for l in n.traverse_postorder(internal=False)])
        else:
            stack.extend(reversed(n.children))
    reps = [(refs[k], [k]) for k in single]
    for g in clusters:
        reps.append((consensus(g, refs, prot), g))
    return reps


def reps_to_json(reps):
    return [{'cons': r[0].tobytes().decode(), 'members': list(r[1])} for r in reps]


def g2():
    tree, n2n, _ = prepare_tree(os.path.join(DATA, 'backbone.nwk'))
    refs = ref_fasta2dic(os.path.join(DATA, 'ref.fa'), False, False)
    qs = ref_fasta2dic(os.path.join(DATA, 'query.fa'), False, False)
    out = {'cases': []}
    reps_cl = clade_clusters(tree, refs, False, 3, 12)
    out['clade_clusters'] = reps_to_json(reps_cl)
    for label, reps in (('singleton', None), ('clades', reps_cl)):
        for f, b in ((0.2, 25), (0.3, 5), (0.15, 60), (10.0, 1)):
            r = make_reference(refs, False, f, b, reps)
            for qn in list(qs)[:4]:
                obs = r.get_obs_dist(qs[qn], qn, 0.001)
                out['cases'].append({'clusters': label, 'f': f, 'b': b, 'query': qn,
                                     'obs': [[k, float(v)] for k, v in obs.items()]})
    jdump(out, 'g2_selection.json')


# ============================================================================= G3 per-edge
def g3():
    tree, n2n, _ = prepare_tree(os.path.join(DATA, 'backbone.nwk'))
    refs = ref_fasta2dic(os.path.join(DATA, 'ref.fa'), False, False)
    qs = ref_fasta2dic(os.path.join(DATA, 'query.fa'), False, False)
    arrays = {}
    for qi, (f, b) in zip(range(3), ((0.2, 25), (0.35, 25), (1e9, 25))):
        qn = list(qs)[qi]
        r = make_reference(refs, False, f, b)
        obs = r.get_obs_dist(qs[qn], qn, 0.001)
        arrays['q%d_obs_names' % qi] = np.array(list(obs))
        arrays['q%d_obs_dist' % qi] = np.array(list(obs.values()), dtype=np.float64)
        for m in ALG:
            st = Subtree(obs, n2n)
            alg = ALG[m](st)
            alg.dp_frag()
            alg.placement_per_edge(False)
            valids = [n for n in st.traverse_postorder() if n.valid]
            sn, rn = TUPLES[m]
            key = 'q%d_%s_' % (qi, m)
            arrays[key + 'edge'] = np.array([n.edge_index for n in valids])
            arrays[key + 'S'] = np.array([[float(getattr(n, a)) for a in sn] for n in valids])
            arrays[key + 'R'] = np.array([[float(getattr(n, a)) for a in rn] for n in valids])
            arrays[key + 'x'] = np.array([[float(n.x_1), float(n.x_2), float(n.x_1_neg), float(n.x_2_neg)]
                                          for n in valids])
            arrays[key + 'err'] = np.array([float(alg.error_per_edge(n)) for n in valids])
            arrays[key + 'lca'] = np.array(st.root.edge_index)
            arrays[key + 'num_nodes'] = np.array(st.num_nodes)
            st.unroll_changes()
            assert not any(n.valid for n in tree.traverse_postorder())
    np.savez_compressed(os.path.join(HERE, 'g3_per_edge.npz'), **arrays)


# ============================================================================= G4 placements
def read_dismat(path):
    import re
    with open(path) as f:
        tags = list(re.split(r'\s+', f.readline().rstrip()))[1:]
        for line in f.readlines():
            d = list(re.split(r'\s+', line.strip()))
            yield (d[0], None, dict(zip(tags, map(float, d[1:]))))


def g4():
    tree, n2n, newick = prepare_tree(os.path.join(DATA, 'backbone.nwk'))
    refs = ref_fasta2dic(os.path.join(DATA, 'ref.fa'), False, False)
    qs = ref_fasta2dic(os.path.join(DATA, 'query.fa'), False, False)
    out = {'tree': newick, 'aln': [], 'dist': [], 'small': [], 'edge_cases': {}}
    for m in ('OLS', 'FM', 'BME', 'BE', 'XYZ'):
        for c in ('MLSE', 'ME', 'HYBRID'):
            for neg in (False, True):
                if m == 'XYZ' and (c != 'MLSE' or neg):
                    continue
                res = run_queries(make_reference(refs, False, 0.2, 25), options(m, c, neg), n2n,
                                  [(k, v, None) for k, v in qs.items()])
                out['aln'].append({'m': m, 'c': c, 'n': neg, 'f': 0.2, 'b': 25, 'p': placements_of(res)})
    for f, b in ((0.35, 10), (1e9, 25)):
        for m in ('OLS', 'FM', 'BME', 'BE'):
            res = run_queries(make_reference(refs, False, f, b), options(m, 'MLSE', False, f, b), n2n,
                              [(k, v, None) for k, v in qs.items()])
            out['aln'].append({'m': m, 'c': 'MLSE', 'n': False, 'f': f, 'b': b, 'p': placements_of(res)})
    # clustered reference (hand-made clades)
    reps_cl = clade_clusters(tree, refs, False, 3, 12)
    for m in ('OLS', 'FM'):
        res = run_queries(make_reference(refs, False, 0.2, 25, reps_cl), options(m), n2n,
                          [(k, v, None) for k, v in qs.items()])
        out['aln'].append({'m': m, 'c': 'MLSE', 'n': False, 'f': 0.2, 'b': 25, 'clusters': 'clades',
                           'p': placements_of(res)})
    # -d on data/dist.mat
    for m in ('OLS', 'FM', 'BME', 'BE'):
        for f, b in ((0.2, 25), (0.3, 5)):
            res = run_queries(None, options(m, 'MLSE', False, f, b), n2n, list(read_dismat(os.path.join(DATA, 'dist.mat'))))
            out['dist'].append({'m': m, 'f': f, 'b': b, 'p': placements_of(res)})
    # -d on the 5-leaf example
    stree, sn2n, snewick = prepare_tree(os.path.join(DATA, 'small_backbone.nwk'))
    out['small_tree'] = snewick
    for m in ('OLS', 'FM', 'BME', 'BE'):
        res = run_queries(None, options(m), sn2n, list(read_dismat(os.path.join(DATA, 'small_dist.mat'))))
        out['small'].append({'m': m, 'p': placements_of(res)})
    # edge cases: all-gap first, name collision, exact duplicate, all-gap later
    rn = list(refs)
    L = len(refs[rn[0]])
    ec = {
        'allgap': np.frombuffer(b'-' * L, dtype='S1'),
        rn[0]: refs[rn[0]],
        'copy_of_second': refs[rn[1]],
        'allgap2': np.frombuffer(b'-' * L, dtype='S1'),
        'normal': qs[list(qs)[0]],
    }
    res = run_queries(make_reference(refs, False, 0.2, 25), options('OLS'), n2n, [(k, v, None) for k, v in ec.items()])
    import copy
    joined = ref_jutil.join_jplace(copy.deepcopy(res))
    out['edge_cases'] = {'names': list(ec), 'results': placements_of(res),
                         'joined': [{'n': p['n'][0], 'p': [x if isinstance(x, int) else float(x) for x in p['p'][0]]}
                                    for p in joined['placements']]}
    # --exclude on ME (int-zero pendant rows become -1)
    res = run_queries(make_reference(refs, False, 0.2, 25), options('OLS', 'ME', False, exclude=True), n2n,
                      [(k, v, None) for k, v in qs.items()])
    out['exclude_ME'] = placements_of(res)
    jdump(out, 'g4_placements.json')


# ============================================================================= G5 tree strings
def g5():
    import re
    j = json.load(open(os.path.join(DATA, 'prot', 'out.jplace')))
    s = j['tree']
    plain = re.sub(r'\{\d+\}', '', s)
    tree, _, newick = prepare_tree(plain, is_text=True)
    assert newick == s, 'reference index_edges + extended_newick must reproduce data/prot/out.jplace'
    out = {'prot_out_sha256': hashlib.sha256(s.encode()).hexdigest(), 'prot_len': len(s)}
    for name in ('small_backbone.nwk', 'backbone.nwk'):
        _, _, nw = prepare_tree(os.path.join(DATA, name))
        out[name] = nw
    _, _, nw = prepare_tree(os.path.join(DATA, 'prot', 'backbone.nwk'))
    out['prot_backbone_sha256'] = hashlib.sha256(nw.encode()).hexdigest()
    jdump(out, 'g5_tree_strings.json')


# ============================================================================= G6 mid-size synthetic
def second_best(st, alg):
    errs = sorted(float(alg.error_per_edge(n)) for n in st.traverse_postorder() if n.valid)
    return errs[0], errs[1]


def g6():
    out = {}
    for label, prot, m in (('nt_OLS', False, 'OLS'), ('aa_FM', True, 'FM')):
        d = synth.make_dataset(2000, 500, 64, protein=prot)
        tree, n2n, _ = prepare_tree(d.newick, is_text=True)
        refs = {n: d.ref_seqs[i].view('S1') for i, n in enumerate(d.ref_names)}
        queries = [(n, d.query_seqs[i].view('S1'), None) for i, n in enumerate(d.query_names)]
        f = 0.6 if prot else 0.2
        ref = make_reference(refs, prot, f, 25)
        opts = options(m, 'MLSE', False, f, 25)
        res = run_queries(ref, opts, n2n, queries)
        gaps, nobs = [], []
        for (qn, qseq, _), r in zip(queries, res):
            obs = ref.get_obs_dist(qseq, qn, 0.001)
            nobs.append(len(obs))
            if r['placements'][0]['p'][0][0] < 0 or len(obs) < 3 or any(v == 0 for v in obs.values()):
                gaps.append(None)
                continue
            st = Subtree(obs, n2n)
            alg = ALG[m](st)
            alg.dp_frag()
            alg.placement_per_edge(False)
            gaps.append(list(second_best(st, alg)))
            st.unroll_changes()
        out[label] = {'N': 2000, 'L': 500, 'Q': 64, 'protein': prot, 'm': m, 'f': f, 'b': 25,
                      'p': placements_of(res), 'best_second': gaps, 'n_obs': nobs}
    # -d / BME from noisy true distances
    d = synth.make_dataset(2000, 500, 64)
    tree, n2n, _ = prepare_tree(d.newick, is_text=True)
    D = synth.noisy_distance_rows(d.tree, d.query_leaf, d.query_pendant, list(range(64)))
    queries = [(d.query_names[i], None, dict(zip(d.ref_names, D[i].tolist()))) for i in range(64)]
    for m in ('BME', 'OLS'):
        res = run_queries(None, options(m, 'MLSE', False, 0.2, 25), n2n, queries)
        out['dmat_' + m] = {'N': 2000, 'Q': 64, 'm': m, 'f': 0.2, 'b': 25, 'p': placements_of(res)}
    jdump(out, 'g6_synthetic.json')


# ============================================================================= G7 whole-CLI runs
def superset_alignment(path):
    """data/ref.fa plus the first three records of data/query.fa: a reference alignment that holds
    rows beyond the backbone's leaves (written at generation time and again by the test)."""
    with open(path, 'w') as f:
        f.write(open(os.path.join(DATA, 'ref.fa')).read())
        recs = open(os.path.join(DATA, 'query.fa')).read().split('>')[1:4]
        f.write(''.join('>' + r for r in recs))
    return path


def g8():
    """-s with rows that are not backbone leaves + -x: the reference ignores those rows as references
    (TreeCluster's table names tree leaves only) but keeps them out of the query set."""
    import tempfile
    tmp = tempfile.mkdtemp()
    sup = superset_alignment(os.path.join(tmp, 'superset_ref.fa'))
    ext = os.path.join(tmp, 'extended_ref.fa')  # = data/ref.fa + data/query.fa
    with open(ext, 'w') as f:
        f.write(open(os.path.join(DATA, 'ref.fa')).read() + open(os.path.join(DATA, 'query.fa')).read())
    g7({'aln_superset': ['-s', sup, '-x', ext, '-t', os.path.join(DATA, 'backbone.nwk'), '-m', 'OLS', '-D', '-T', '2']},
       rename={sup: 'superset_ref.fa', ext: 'extended_ref.fa'})


def g9():
    """Branch lengths of the FastTree binary the reference bundles (apples/tools/FastTree-linux, FastTree 2.1.11)
    run as the reference runs it (apples/reestimateBackbone.py:82-84): pins oracle/fasttree_me.py and the HIP
    estimator.  Fixtures: the binary's Newick output for data/backbone.nwk + data/ref.fa, for a rooted synthetic
    nucleotide set with odd symbols and heavy gaps, and for a rooted synthetic protein set (inputs are
    regenerated from seeds by the tests)."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.dirname(HERE))
    from fasttree_cases import fasttree_case, fasttree_cases
    ft = os.path.join(REF, 'apples', 'tools', 'FastTree-linux')
    tmp = tempfile.mkdtemp()

    def run(tree_fp, fasta_fp, protein, out_name):
        cmd = [ft, '-nosupport', '-nome', '-noml', '-intree', tree_fp] + ([] if protein else ['-nt'])
        with open(fasta_fp) as f:
            p = subprocess.run(cmd, stdin=f, capture_output=True, check=True)
        with open(os.path.join(HERE, out_name), 'w') as o:
            o.write(p.stdout.decode().strip() + '\n')

    run(os.path.join(DATA, 'backbone.nwk'), os.path.join(DATA, 'ref.fa'), False, 'g9_fasttree_data.nwk')
    for name, (n, L, protein, seed, odd, mean_len) in fasttree_cases().items():
        d, seqs = fasttree_case(n, L, protein, seed, odd, mean_len)
        tfp, ffp = os.path.join(tmp, name + '.nwk'), os.path.join(tmp, name + '.fa')
        open(tfp, 'w').write(d.newick + '\n')
        with open(ffp, 'w') as f:
            for nm, sq in zip(d.ref_names, seqs):
                f.write('>%s\n%s\n' % (nm, bytes(sq).decode()))
        run(tfp, ffp, protein, 'g9_fasttree_%s.nwk' % name)


# ============================================================================= G10 the default protein route
def g10():
    """-p with clusters (apples/Reference.py:117-157 over scoredist, apples/PoolRepresentativeWorker.py:33-58 with the
    21-symbol alphabet): consensus rows, observed dicts in order, placements, and one whole run of the reference's
    run_apples.py -p consuming this repo's cluster table.  Inputs: tests/prot_cases.py (seeded; the tests regenerate them)."""
    import tempfile
    sys.path.insert(0, os.path.dirname(HERE))
    import prot_cases
    tmp = tempfile.mkdtemp()
    ref_fp, qry_fp, tree_fp = prot_cases.write_case(tmp)
    tree, n2n, newick = prepare_tree(tree_fp)
    refs = ref_fasta2dic(ref_fp, True, False)
    qs = ref_fasta2dic(qry_fp, True, False)
    reps_cl = clade_clusters(tree, refs, True, prot_cases.CLADE_MIN, prot_cases.CLADE_MAX)
    out = {'clade_clusters': reps_to_json(reps_cl), 'cases': [], 'placements': [], 'tree': newick}
    for label, reps in (('singleton', None), ('clades', reps_cl)):
        for f, b in prot_cases.SELECTION_PARAMS:
            r = make_reference(refs, True, f, b, reps)
            for qn in list(qs)[:6]:
                obs = r.get_obs_dist(qs[qn], qn, 0.001)
                out['cases'].append({'clusters': label, 'f': f, 'b': b, 'query': qn,
                                     'obs': [[k, float(v)] for k, v in obs.items()]})
    queries = [(k, v, None) for k, v in qs.items()]
    for m, c, neg in (('FM', 'MLSE', False), ('OLS', 'MLSE', False), ('BME', 'MLSE', False), ('BE', 'MLSE', False),
                      ('FM', 'ME', False), ('FM', 'HYBRID', False), ('FM', 'MLSE', True), ('OLS', 'HYBRID', True)):
        for f, b in prot_cases.SELECTION_PARAMS[:2]:
            res = run_queries(make_reference(refs, True, f, b, reps_cl), options(m, c, neg, f, b), n2n, queries)
            out['placements'].append({'m': m, 'c': c, 'n': neg, 'f': f, 'b': b, 'clusters': 'clades', 'p': placements_of(res)})
    with open(os.path.join(HERE, 'g10_prot_clustered.json'), 'w') as f:  # (one case per line: the observed dicts are long)
        f.write('{' + ',\n'.join('%s: %s' % (json.dumps(k), json.dumps(v) if not isinstance(v, list) else
                                             '[\n' + ',\n'.join(json.dumps(x) for x in v) + '\n]') for k, v in out.items()) + '}\n')
    g7({'prot_default': ['-p', '-s', ref_fp, '-q', qry_fp, '-t', tree_fp, '-D', '-T', '2'],
        'prot_OLS_f01_b5': ['-p', '-s', ref_fp, '-q', qry_fp, '-t', tree_fp, '-m', 'OLS', '-f', '0.1', '-b', '5', '-D', '-T', '2']},
       rename={ref_fp: 'prot_ref.fa', qry_fp: 'prot_query.fa', tree_fp: 'prot_backbone.nwk'})


def g7(runs=None, rename=None):
    """Run the reference's own run_apples.py end to end (treeswift stand-in registered above, and
    a stub TreeCluster.py on PATH that labels every leaf '-1' = all-singleton clusters)."""
    import runpy
    import stat
    import tempfile
    tmp = tempfile.mkdtemp()
    stub = os.path.join(tmp, 'TreeCluster.py')
    # The reference shells out to TreeCluster.py (apples/Reference.py:87-88), which is not installed.
    # The stand-in writes the table of this repo's own max-diameter clustering for the requested
    # threshold (apples_amd/treecluster.py), or all singletons when APPLES_STUB_SINGLETONS is set, so
    # the reference's cluster handling (consensus, heap expansion) runs on exactly the clusters the
    # build uses.
    with open(stub, 'w') as f:
        f.write('#!%s\nimport os, sys\nsys.path.insert(0, %r)\nfrom apples_amd.tree import read_tree\n'
                'from apples_amd import treecluster\n'
                'a = sys.argv\nt = read_tree(a[a.index("-i") + 1])\nout = a[a.index("-o") + 1]\n'
                'if os.environ.get("APPLES_STUB_SINGLETONS"):\n'
                '    f = open(out, "w"); f.write("SequenceName\\tClusterNumber\\n")\n'
                '    [f.write("%%s\\t-1\\n" %% t.labels[v]) for v in t.leaves]; f.close()\n'
                'else:\n'
                '    treecluster.write_table(t, float(a[a.index("-t") + 1]), out)\n' % (sys.executable, ROOT))
    os.chmod(stub, os.stat(stub).st_mode | stat.S_IEXEC)
    os.environ['PATH'] = tmp + os.pathsep + os.environ['PATH']
    runs = runs or {
        'aln_OLS': ['-s', os.path.join(DATA, 'ref.fa'), '-q', os.path.join(DATA, 'query.fa'), '-t',
                    os.path.join(DATA, 'backbone.nwk'), '-m', 'OLS', '-D', '-T', '2'],
        'aln_default': ['-s', os.path.join(DATA, 'ref.fa'), '-q', os.path.join(DATA, 'query.fa'), '-t',
                        os.path.join(DATA, 'backbone.nwk'), '-D', '-T', '2'],
        'aln_f03_b5_BME': ['-s', os.path.join(DATA, 'ref.fa'), '-q', os.path.join(DATA, 'query.fa'), '-t',
                           os.path.join(DATA, 'backbone.nwk'), '-m', 'BME', '-f', '0.3', '-b', '5', '-D', '-T', '2'],
        'aln_OLS_singletons': ['-s', os.path.join(DATA, 'ref.fa'), '-q', os.path.join(DATA, 'query.fa'), '-t',
                               os.path.join(DATA, 'backbone.nwk'), '-m', 'OLS', '-D', '-T', '2'],
        'dist_default': ['-d', os.path.join(DATA, 'dist.mat'), '-t', os.path.join(DATA, 'backbone.nwk'), '-T', '2'],
        'small_BME': ['-d', os.path.join(DATA, 'small_dist.mat'), '-t', os.path.join(DATA, 'small_backbone.nwk'),
                      '-m', 'BME', '-T', '1'],
    }
    for label, args in runs.items():
        outp = os.path.join(HERE, 'g7_cli_%s.jplace' % label)
        if label.endswith('_singletons'):
            os.environ['APPLES_STUB_SINGLETONS'] = '1'
        else:
            os.environ.pop('APPLES_STUB_SINGLETONS', None)
        old = sys.argv
        sys.argv = ['run_apples.py'] + args + ['-o', outp]
        try:
            runpy.run_path(os.path.join(REF, 'run_apples.py'), run_name='__main__')
        except RuntimeError as e:  # set_start_method may only be called once per process
            if 'context has already been set' not in str(e):
                raise
            import multiprocessing as mp
            orig = mp.set_start_method
            mp.set_start_method = lambda *a, **k: None
            try:
                runpy.run_path(os.path.join(REF, 'run_apples.py'), run_name='__main__')
            finally:
                mp.set_start_method = orig
        finally:
            sys.argv = old
        # keep the fixture location-independent
        j = json.load(open(outp))
        j['metadata']['invocation'] = 'run_apples.py ' + ' '.join(
            (rename or {}).get(a, a).replace(DATA + os.sep, 'data/') for a in args)
        with open(outp, 'w') as f:
            f.write(json.dumps(j, sort_keys=True, indent=4) + '\n')


if __name__ == '__main__':
    which = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g5', 'g6', 'g7', 'g8', 'g9', 'g10']
    for w in which:
        print('generating', w, flush=True)
        globals()[w]()
    print('done')
