"""HIP path vs the C oracle at BASELINE config C2's reference size, plus size-independent
properties at sizes the oracle would take too long for.  Needs an MI355X."""
import os
import sys

import numpy as np
import pytest

from helpers import ROOT

sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from oracle_c import COracle  # noqa: E402

from apples_amd import synth  # noqa: E402
from apples_amd.engine import Engine, jc69_lut, F_EXACT, F_INSUFFICIENT  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def c2():
    d = synth.make_dataset(10000, 1000, 1536)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    return d, nodes


@pytest.fixture(scope='module')
def c2_full():
    """BASELINE config 2 in full: 10 000 leaves, L = 1000, 10 000 queries."""
    d = synth.make_dataset(10000, 1000, 10000)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    return d, nodes


def _compare(got, want, co, d, nodes, what):
    """Edges bit-exact, lengths within 1e-6 relative; a differing edge is accepted only if it is the
    documented tie class: the two candidate edges' residuals agree to 1e-12."""
    assert np.array_equal(got['flags'] & (F_EXACT | F_INSUFFICIENT), want['flags'] & (F_EXACT | F_INSUFFICIENT)), what
    assert np.array_equal(got['n_obs'], want['n_obs']), what
    assert np.array_equal(got['n_valid'], want['n_valid']), what
    diff = np.nonzero(got['edge'] != want['edge'])[0]
    assert len(diff) == 0, '%s: %d edge mismatches' % (what, len(diff))
    # same integers, same distance table, same IEEE operations, libm's pow bits: everything is bit-identical
    assert got.tobytes() == want.tobytes(), what
    return 0


@pytest.mark.parametrize('method,criterion', [('OLS', 'MLSE'), ('FM', 'MLSE'), ('BME', 'HYBRID'), ('BE', 'ME')])
def test_c2_reference_size_against_c_oracle(c2, method, criterion):
    d, nodes = c2
    nthreads = len(os.sched_getaffinity(0))
    co = COracle(d.tree, d.ref_seqs, nodes, method=method, criterion=criterion, lut=jc69_lut(1000, 0.001),
                 threads=nthreads)
    nq = 1536 if method == 'OLS' else 384
    want = co.place_sequences(d.query_seqs[:nq])
    eng = Engine(d.tree, d.ref_seqs, nodes, method=method, criterion=criterion)
    got = eng.place_sequences(d.query_seqs[:nq])
    _compare(got, want, co, d, nodes, 'c2 %s/%s' % (method, criterion))
    eng.close()


@pytest.mark.parametrize('layout', ['scan', 'bits', 'map', 'merge', 'lean'])
def test_all_observed_stress_and_batching(c2, layout, monkeypatch):
    """-f huge: every leaf observed (worst case for both kernels, V = 2N-2; more than 4 096 observed
    leaves send every query to the workgroup-sized teams); also forces several device batches and
    checks they agree with one big batch.  Both node-lookup layouts of the level-loop sweep and the scan
    formulation (per-leaf state beyond a team's LDS share)."""
    d, nodes = c2
    if layout == 'scan':
        monkeypatch.setenv('APPLES_SWEEP_SCAN', '1')
    if layout == 'map':
        monkeypatch.setenv('APPLES_NODE_MAP', '1')
    if layout == 'bits':  # (a binary tree of 2 048 nodes or more takes the lean form by default)
        monkeypatch.setenv('APPLES_NO_SWEEP_MERGE', '1')
    if layout in ('merge', 'lean'):  # merged level lists inside the level loop, or (binary trees, the default from 2 048
        monkeypatch.setenv('APPLES_SWEEP_MERGE', '1')  # nodes up) the three-pass lean form
    if layout == 'merge':
        monkeypatch.setenv('APPLES_NO_SWEEP_LEAN', '1')
    nthreads = len(os.sched_getaffinity(0))
    co = COracle(d.tree, d.ref_seqs, nodes, method='OLS', threshold=1e9, lut=jc69_lut(1000, 0.001), threads=nthreads)
    want = co.place_sequences(d.query_seqs[:96])
    eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS', threshold=1e9, max_batch=32)
    got = eng.place_sequences(d.query_seqs[:96])
    assert eng.describe()['batch'] == 32 and eng.describe()['sweep'] == ('scan' if layout == 'scan' else 'levels')
    assert eng.describe()['sweep_layout'] == layout
    _compare(got, want, co, d, nodes, 'all observed')
    assert (got['n_valid'] >= 2 * 10000 - 3).all()
    eng.close()


def test_big_tree_node_map_and_tag_wrap_around(c2, monkeypatch):
    """Trees of this size keep the sweep's valid-node bits in LDS; big trees use a tagged node map
    in global scratch that is never cleared between queries (the table is wiped only when the
    tags run out).  Force that layout here, with 3-bit tags and 8 teams so that every team wraps
    dozens of times; placements must not change."""
    d, nodes = c2
    eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS')
    want = eng.place_sequences(d.query_seqs)
    eng.close()
    monkeypatch.setenv('APPLES_NODE_MAP', '1')
    monkeypatch.setenv('APPLES_MAP_BITS', '29')
    monkeypatch.setenv('APPLES_SWEEP_TEAMS', '8')
    eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS')
    got = eng.place_sequences(d.query_seqs)
    again = eng.place_sequences(d.query_seqs)
    eng.close()
    assert got.tobytes() == want.tobytes()
    assert again.tobytes() == want.tobytes()


def test_properties_at_full_c2_query_count(c2_full):
    """Size-independent properties on the whole C2 pass (all 10 000 queries): determinism across runs
    and batch sizes, permutation equivariance over queries, duplicates of reference rows place exactly;
    a strided sample of 80 queries byte for byte against the C oracle."""
    d, nodes = c2_full
    eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS')
    q = d.query_seqs
    assert len(q) == 10000
    a = eng.place_sequences(q)
    sample = np.arange(0, 10000, 125)
    co = COracle(d.tree, d.ref_seqs, nodes, method='OLS', lut=jc69_lut(1000, 0.001), threads=len(os.sched_getaffinity(0)))
    assert a[sample].tobytes() == co.place_sequences(q[sample]).tobytes()
    b = eng.place_sequences(q)
    assert a.tobytes() == b.tobytes()
    perm = np.random.default_rng(0).permutation(len(q))
    c = eng.place_sequences(q[perm])
    assert c.tobytes() == a[perm].tobytes()
    e2 = Engine(d.tree, d.ref_seqs, nodes, method='OLS', max_batch=160)
    assert e2.place_sequences(q).tobytes() == a.tobytes()
    e2.close()
    # a query identical to reference row r has distance 0 to it -> exact placement on that leaf's edge
    rows = np.arange(0, 10000, 97)
    ex = eng.place_sequences(d.ref_seqs[rows])
    assert (ex['flags'] & F_EXACT).all()
    # identical reference rows can exist; the placed leaf must be one at distance zero
    _, dist = eng.distances(d.ref_seqs[rows[:8]], want_counts=False)
    for k in range(8):
        placed_row = int(np.nonzero(nodes == ex['edge'][k])[0][0])
        assert dist[k, placed_row] == 0
    eng.close()


def test_fused_and_full_row_selection_paths_agree(c2):
    """The fused path (threshold compaction in the distance epilogue + listed top-up) and the
    full-row path (APPLES_NO_FUSE=1) are two routes to the same observed sets: placements must be
    byte-identical, also when every query needs the top-up rule (tiny threshold) and when large
    queries are routed to workgroup-sized sweep teams.  The top-up rule itself has two forms
    (streaming the full row, or ranking per-segment minima, the default for long rows): forced
    here with APPLES_TOPUP_MIN_ROWS=0.  And the pair counts come from the fp4 matrix-core kernel
    by default, from the bit-plane VALU kernel with APPLES_NO_DIST_MFMA=1."""
    import subprocess
    d, nodes = c2
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from apples_amd import synth\n"
            "from apples_amd.engine import Engine\n"
            "d = synth.make_dataset(10000, 1000, 1536)\n"
            "nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)\n"
            "out = []\n"
            "for thr, b in ((0.2, 25), (0.01, 25), (0.35, 40), (0.02, 200)):\n"
            "    e = Engine(d.tree, d.ref_seqs, nodes, method='OLS', threshold=thr, baseobs=b)\n"
            "    out.append(e.place_sequences(d.query_seqs[:512]).tobytes()); e.close()\n"
            "sys.stdout.buffer.write(b''.join(out))\n" % ROOT)
    outs = []
    for env in ({}, {'APPLES_NO_FUSE': '1'}, {'APPLES_BIG_THRESHOLD': '300'}, {'APPLES_SWEEP_TEAM': '256'},
                {'APPLES_TOPUP_MIN_ROWS': '0'}, {'APPLES_NO_DIST_MFMA': '1'}, {'APPLES_NO_DIST_GEMM': '1'}, {'APPLES_GEMM_TABLE': '1'}, {'APPLES_GEMM_QT': '128'}, {'APPLES_GEMM_QT': '128', 'APPLES_GEMM_TABLE': '1'}, {'APPLES_SWEEP_SCAN': '1'},
                {'APPLES_SWEEP_MERGE': '1'}, {'APPLES_SWEEP_MERGE': '1', 'APPLES_BIG_THRESHOLD': '300'}, {'APPLES_SWEEP_MERGE': '1', 'APPLES_SWEEP_TEAM': '256'},
                {'APPLES_SWEEP_MERGE': '1', 'APPLES_NO_SWEEP_LEAN': '1'}, {'APPLES_SWEEP_MERGE': '1', 'APPLES_NO_SWEEP_LEAN': '1', 'APPLES_BIG_THRESHOLD': '300'},
                {'APPLES_SWEEP_MERGE': '1', 'APPLES_LEAN_UP_WGS': '2', 'APPLES_LEAN_DOWN_WGS': '3'}, {'APPLES_SWEEP_MERGE': '1', 'APPLES_LEAN_POOL_MB': '1'},
                {'APPLES_SWEEP_MERGE': '1', 'APPLES_LEAN_BIG_TEAM': '512', 'APPLES_BIG_THRESHOLD': '300'},
                {'APPLES_SWEEP_MERGE': '1', 'APPLES_NO_SWEEP_LEAN': '1', 'APPLES_SWEEP_CAP': '600'},
                {'APPLES_SWEEP_SCAN': '1', 'APPLES_BIG_THRESHOLD': '300'}, {'APPLES_SWEEP_SCAN': '1', 'APPLES_SWEEP_TEAM': '256'}):
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, env=dict(os.environ, **env), timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(r.stdout)
    assert len(outs[0]) == 4 * 512 * 40
    assert all(o == outs[0] for o in outs)


def test_debug_switches_cross_the_routes_inside_one_process(c2):
    """apples_params.debug (include/apples_hip.h APPLES_DBG_*) selects, per context, the alternative routes that the
    environment knobs select per process: full-row selection, the layouts of the level-loop sweep, the lean and the scan
    sweep, the bit-plane-fed distance pass.  One process, one engine per combination, identical bytes."""
    d, nodes = c2
    outs = {}
    for dbg in ((), ('no_fuse',), ('node_map',), ('sweep_merge',), ('sweep_merge', 'no_sweep_lean'), ('no_dist_gemm',), ('sweep_scan',),
                ('no_fuse', 'sweep_merge'), ('no_dist_gemm', 'node_map'), ('no_sweep_merge',), ('no_sweep_lean',)):
        eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS', debug=dbg)
        info = eng.describe()
        if 'sweep_scan' in dbg:
            assert info['sweep'] == 'scan'
        elif 'sweep_merge' in dbg:
            assert info['sweep_layout'] == ('merge' if 'no_sweep_lean' in dbg else 'lean')
        elif 'node_map' in dbg:
            assert info['sweep_layout'] == 'map'
        elif 'no_sweep_merge' in dbg or 'no_sweep_lean' in dbg:
            assert info['sweep_layout'] == 'bits'  # (the node bits of 20 000 nodes fit in LDS)
        else:
            assert info['sweep_layout'] == 'lean'  # binary tree, 2 048 nodes or more
        assert ('gemm' in info['fused_distance_pass']) == ('no_dist_gemm' not in dbg)
        outs[dbg] = eng.place_sequences(d.query_seqs[:768]).tobytes()
        eng.close()
    assert len(set(outs.values())) == 1, [k for k, v in outs.items() if v != outs[()]]


def test_device_resident_results_visible_to_torch_zero_copy():
    """bench.py hands the device-resident placement structs to torch.distributed (RCCL gather)
    through __cuda_array_interface__ without a host round trip: the view must alias the bytes that
    apples_fetch_placements copies out.  Runs in a fresh process, as bench.py does."""
    import subprocess
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
            "from apples_amd import synth\n"
            "from apples_amd.engine import Engine\n"
            "from apples_amd.distributed import shard_bounds\n"
            "d = synth.make_dataset(2000, 500, 256)\n"
            "nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)\n"
            "eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS')\n"
            "h, nq = eng.upload_queries(d.query_seqs)\n"
            "eng.place_resident(h)\n"
            "class V:\n"
            "    def __init__(self, ptr, n):\n"
            "        self.__cuda_array_interface__ = {'shape': (n,), 'typestr': '|u1', 'data': (ptr, False), 'version': 2}\n"
            "view = torch.as_tensor(V(eng.placements_device_ptr(h), nq * 40), device='cuda')\n"
            "want = eng.fetch(h, nq)\n"
            "assert view.cpu().numpy().tobytes() == want.tobytes()\n"
            "assert shard_bounds(nq, 1) == [(0, nq)]\n"
            "print('ZEROCOPY_OK')\n" % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=900)
    assert 'ZEROCOPY_OK' in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_resident_distance_table_matches_host_entry_point():
    """apples_table_upload + apples_place_resident (the bench's C5 path) against
    apples_place_from_distances and the C oracle on a 2000-leaf table."""
    d = synth.make_dataset(2000, 8, 96)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    ix = synth.TreeIndex(d.tree)
    D = synth.fast_distance_rows(d.tree, ix, d.query_leaf, d.query_pendant, list(range(96)))
    D[5, ::3] = -1.0           # missing values
    D[6, :] = -1.0             # nothing observed
    D[7, 10] = 0.0             # exact hit
    for m in ('BME', 'FM'):
        eng = Engine(d.tree, None, method=m)
        a = eng.place_distances(D, nodes)
        h, n = eng.upload_table(D, nodes)
        eng.place_resident(h)
        b = eng.fetch(h, n)
        assert a.tobytes() == b.tobytes()
        want = COracle(d.tree, method=m).place_distances(D, nodes)
        _compare(b, want, None, d, nodes, 'table %s' % m)
        assert b[6]['flags'] & F_INSUFFICIENT and b[7]['flags'] & F_EXACT
        eng.free_queries(h)
        eng.close()


def test_big_backbone_sample_against_c_oracle():
    """A 100 k-leaf backbone exercises what the 10 k-leaf tests cannot: the tagged node map of the
    sweep (the bit space no longer fits LDS), the top-up selection by segment minima (rows of 40 k
    and more), multi-megabyte observation lists.  Thousands of queries go through the device in one
    pass; a sample of them, including the ones with the fewest and the most observed leaves, is
    checked byte for byte against the C oracle."""
    d = synth.make_dataset(100000, 500, 2048)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS')
    info = eng.describe()
    assert info['n_nodes'] == 199999
    got = eng.place_sequences(d.query_seqs)
    eng.close()
    order = np.argsort(got['n_obs'], kind='stable')
    sample = np.unique(np.concatenate([order[:6], order[-6:], np.arange(0, 2048, 171)]))
    co = COracle(d.tree, d.ref_seqs, nodes, method='OLS', lut=jc69_lut(500, 0.001), threads=len(os.sched_getaffinity(0)))
    want = co.place_sequences(d.query_seqs[sample])
    assert got[sample].tobytes() == want.tobytes()
    assert (got['n_obs'] >= 25).all() or (got['flags'] & (F_EXACT | F_INSUFFICIENT)).any()


def test_big_protein_backbone_sample_against_c_oracle():
    """The same at the amino-acid benchmark shape (50 k leaves, scoredist + FM): edges and counts
    identical, lengths within 1e-9 relative (the table sums are fp64 on both sides; only their order
    of summation is not pinned by the reference)."""
    d = synth.make_dataset(50000, 300, 1024, protein=True)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    eng = Engine(d.tree, d.ref_seqs, nodes, protein=True, method='FM')
    got = eng.place_sequences(d.query_seqs)
    eng.close()
    order = np.argsort(got['n_obs'], kind='stable')
    sample = np.unique(np.concatenate([order[:4], order[-4:], np.arange(0, 1024, 128)]))
    co = COracle(d.tree, d.ref_seqs, nodes, protein=True, method='FM', threads=len(os.sched_getaffinity(0)))
    want = co.place_sequences(d.query_seqs[sample])
    g = got[sample]
    for f in ('edge', 'flags', 'n_obs', 'n_valid'):
        assert np.array_equal(g[f], want[f]), f
    for f in ('error', 'distal', 'pendant'):
        np.testing.assert_allclose(g[f], want[f], rtol=1e-9, atol=1e-15, err_msg=f)


def test_big_distance_table_against_c_oracle():
    """-d input at benchmark shape: a 100 k-column table (noisy path distances), BME, every row
    checked byte for byte against the C oracle; rows with missing values, one exact hit."""
    d = synth.make_dataset(100000, 8, 96)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    ix = synth.TreeIndex(d.tree)
    D = synth.fast_distance_rows(d.tree, ix, d.query_leaf, d.query_pendant, list(range(96)))
    D[5, ::3] = -1.0
    D[7, 4711] = 0.0
    eng = Engine(d.tree, None, method='BME')
    h, n = eng.upload_table(D, nodes)
    eng.place_resident(h)
    got = eng.fetch(h, n)
    eng.free_queries(h)
    eng.close()
    want = COracle(d.tree, method='BME', threads=len(os.sched_getaffinity(0))).place_distances(D, nodes)
    assert got.tobytes() == want.tobytes()
    assert got[7]['flags'] & F_EXACT


@pytest.mark.parametrize('method,criterion', [('OLS', 'MLSE'), ('FM', 'HYBRID')])
def test_clustered_reference_at_c2_size_against_c_oracle(c2, method, criterion):
    """The command line's default route: max-diameter clusters at 1.2 x -f, consensus representatives,
    heap-ordered cluster expansion (apples/Reference.py:117-157) -- full distance rows and the general
    selection kernel -- at the 10 k-leaf benchmark size, byte for byte against the C oracle."""
    from apples_amd import treecluster
    from apples_amd.fasta import Alignment
    from apples_amd.reference import ReducedReference
    d, nodes = c2
    ref = ReducedReference(Alignment(d.ref_names, d.ref_seqs), False, treecluster.grouped(d.tree, 0.2 * 1.2))
    ca = ref.cluster_arrays()
    assert len(ca[0]) > 100  # multi-member clusters with consensus rows
    nq = 512
    co = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, method=method, criterion=criterion, lut=jc69_lut(1000, 0.001),
                 threads=len(os.sched_getaffinity(0)))
    want = co.place_sequences(d.query_seqs[:nq])
    eng = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method=method, criterion=criterion)
    got = eng.place_sequences(d.query_seqs[:nq])
    eng.close()
    _compare(got, want, co, d, nodes, 'clustered %s/%s' % (method, criterion))


def test_cluster_major_member_distances_on_awkward_cluster_sizes(c2, monkeypatch):
    """The clustered fast path computes member distances cluster by cluster, a tile of the accepting queries at a time
    (select.hip:k_cluster_dist).  Cluster sizes that exercise its shapes: more than 256 members (member chunks), 65-85
    (three query lanes), singletons, pairs and triples (256, 128 and 85 query lanes), lists that end in partial tiles.
    Against the C oracle byte for byte, and the by-query form of the same kernel (APPLES_CLUSTER_BY_QUERY) likewise."""
    from apples_amd import treecluster
    from apples_amd.fasta import Alignment
    from apples_amd.reference import ReducedReference
    d, nodes = c2
    groups = treecluster.grouped(d.tree, 0.4)
    sizes = sorted(len(g[1]) for g in groups)
    assert sizes[-1] > 512 and sum(1 for s in sizes if s > 256) >= 8
    # the smallest cluster is cut into singletons, pairs and triples
    groups.sort(key=lambda g: len(g[1]))
    small = groups.pop(0)[1]
    i, k = 0, 0
    while i < len(small):
        n = 1 + k % 3
        groups.append(('cut%d' % k, small[i:i + n]))
        i += n
        k += 1
    ref = ReducedReference(Alignment(d.ref_names, d.ref_seqs), False, groups)
    ca = ref.cluster_arrays()
    nq = 700  # (not a multiple of the tile sizes)
    thr = 0.45
    co = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', criterion='MLSE', threshold=thr,
                 lut=jc69_lut(1000, 0.001), threads=len(os.sched_getaffinity(0)))
    want = co.place_sequences(d.query_seqs[:nq])
    for by_query in (False, True):
        if by_query:
            monkeypatch.setenv('APPLES_CLUSTER_BY_QUERY', '1')
        eng = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', criterion='MLSE', threshold=thr)
        assert eng.describe()['cluster_fused'], 'the clustered fast path is not the one under test'
        got = eng.place_sequences(d.query_seqs[:nq])
        eng.close()
        _compare(got, want, co, d, nodes, 'awkward clusters, by query %s' % by_query)
    assert (want['n_obs'] > 300).sum() > nq // 2  # whole big clusters were accepted


@pytest.mark.parametrize('thr,baseobs', [(0.02, 25), (0.05, 200), (0.0, 5)])
def test_clustered_top_up_rule_over_the_representatives(c2, thr, baseobs, monkeypatch):
    """Queries with fewer than `-b` valid member distances inside the threshold (apples/Reference.py:144-152: the heap walk goes
    on beyond the threshold) are served by phase 4 of k_select_clusters: distances to every representative, the next
    smallest (distance, index) until `-b` is reached, the clusters those accept.  A tight threshold sends most queries there.
    Against the C oracle byte for byte, and against the general route (full rows + k_select, APPLES_NO_FUSE)."""
    from apples_amd import treecluster
    from apples_amd.fasta import Alignment
    from apples_amd.reference import ReducedReference
    d, nodes = c2
    ref = ReducedReference(Alignment(d.ref_names, d.ref_seqs), False, treecluster.grouped(d.tree, 0.2 * 1.2))
    ca = ref.cluster_arrays()
    nq = 600
    co = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', criterion='MLSE', threshold=thr, baseobs=baseobs,
                 lut=jc69_lut(1000, 0.001), threads=len(os.sched_getaffinity(0)))
    want = co.place_sequences(d.query_seqs[:nq])
    eng = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', criterion='MLSE', threshold=thr, baseobs=baseobs)
    assert eng.describe()['cluster_fused']
    got = eng.place_sequences(d.query_seqs[:nq])
    eng.close()
    _compare(got, want, co, d, nodes, 'clustered top-up thr %g b %d' % (thr, baseobs))
    monkeypatch.setenv('APPLES_NO_FUSE', '1')
    eng = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', criterion='MLSE', threshold=thr, baseobs=baseobs)
    gen = eng.place_sequences(d.query_seqs[:nq])
    eng.close()
    assert gen.tobytes() == got.tobytes()


@pytest.mark.parametrize('diam', [0.08, 0.005])
def test_queries_that_accept_more_clusters_than_a_workgroups_lists_hold(c2, diam, monkeypatch):
    """1 923 small clusters and a threshold that accepts nearly all of them for every query: more than the 512 the fast
    path's workgroups keep in LDS, so phase 1 lists the queries and the phases' second form (5 120 clusters per workgroup,
    a launch of its own) serves them -- up to 1 024 per device batch; with 1 536 queries in one batch the rest goes to the
    general route.  Against the C oracle byte for byte, and with that second form switched off (APPLES_NO_CLUSTER_BIG); once more
    with more clusters than the second form holds."""
    from apples_amd import treecluster
    from apples_amd.fasta import Alignment
    from apples_amd.reference import ReducedReference
    d, nodes = c2
    ref = ReducedReference(Alignment(d.ref_names, d.ref_seqs), False, treecluster.grouped(d.tree, diam))
    ca = ref.cluster_arrays()
    # (diameter 0.005: nearly every leaf its own cluster, more clusters than the second form holds either -- the general route)
    assert 512 < len(ca[1]) <= 5120 if diam > 0.01 else len(ca[1]) > 5120
    nq = 1536 if diam > 0.01 else 300
    co = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', criterion='MLSE', threshold=1.0,
                 lut=jc69_lut(1000, 0.001), threads=len(os.sched_getaffinity(0)))
    want = co.place_sequences(d.query_seqs[:nq])
    assert (want['n_obs'] > 5000).sum() > nq // 2  # most of the tree observed: far more than 512 clusters accepted
    for off in (False, True):
        if off:
            monkeypatch.setenv('APPLES_NO_CLUSTER_BIG', '1')
        eng = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', criterion='MLSE', threshold=1.0)
        assert eng.describe()['cluster_fused']
        got = eng.place_sequences(d.query_seqs[:nq])
        eng.close()
        _compare(got, want, co, d, nodes, 'more than ACC_CAP clusters, second form off: %s' % off)


@pytest.mark.parametrize('no_fuse', [False, True])
def test_exact_matches_among_the_members_of_one_cluster(no_fuse, monkeypatch):
    """Short branches: many queries are identical to a reference, and some to TWO references of the same cluster -- the
    reference reports the first zero in dict order (PoolQueryWorker.py:73-79: cluster by (d_rep, index), then member
    position).  A differential fuzz (scripts/cluster_fuzz.py) once found the general selection pairing the right member
    position with its neighbour's node (an unrolled loop named element -1 of a register array in an arm that is never
    taken at that iteration, and the optimiser made the rest of the loop pay for it); this is that configuration, against
    the C oracle on both routes."""
    from apples_amd import treecluster
    from apples_amd.fasta import Alignment
    from apples_amd.reference import ReducedReference
    d = synth.make_dataset(60, 950, 879, gap_rate=0.0, seed_tree=203, mean_len=0.003)
    nodes = np.array([d.tree.name_to_node[x] for x in d.ref_names], np.int32)
    ref = ReducedReference(Alignment(d.ref_names, d.ref_seqs), False, treecluster.grouped(d.tree, 0.24))
    ca = ref.cluster_arrays()
    co = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', criterion='MLSE', threshold=0.02, baseobs=25,
                 lut=jc69_lut(950, 0.001), threads=len(os.sched_getaffinity(0)))
    want = co.place_sequences(d.query_seqs)
    assert ((want['flags'] & F_EXACT) != 0).sum() > 50
    if no_fuse:
        monkeypatch.setenv('APPLES_NO_FUSE', '1')
    eng = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', criterion='MLSE', threshold=0.02, baseobs=25, max_batch=512)
    got = eng.place_sequences(d.query_seqs)
    eng.close()
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize('L', [2047, 2050, 3001, 4000, 4092, 4093, 4097, 8190])
def test_long_alignments_through_the_fused_matrix_core_pass(L):
    """The fused distance pass packs (valid, mism) into 13-bit fields and counts in f32
    accumulators over 64-site fp4 blocks: alignments on both sides of the GEMM form's two limits (2 046 sites with the validity
    sum at 2^13, 4 092 at 2^12), just past 4 096 sites (odd number of blocks,
    ragged last word) and just below the 8 192 limit of the packed format, with heavy gaps so that
    the overlap rule and short valid counts occur.  Placements identical to the C oracle's."""
    d = synth.make_dataset(600, L, 300, seed_tree=11, seed_aln=12, seed_query=13)
    rng = np.random.default_rng(L)
    ref = d.ref_seqs.copy()
    qry = d.query_seqs.copy()
    ref[rng.random(ref.shape) < 0.3] = ord('-')           # a third of every row missing
    qry[:, : L // 2][rng.random((len(qry), L // 2)) < 0.5] = ord('-')
    qry[7] = ord('-')                                      # all-gap query
    qry[8] = ref[3]                                        # exact copy of a reference row
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    co = COracle(d.tree, ref, nodes, method='OLS', criterion='MLSE', lut=jc69_lut(L, 0.001),
                 threads=len(os.sched_getaffinity(0)))
    want = co.place_sequences(qry)
    eng = Engine(d.tree, ref, nodes, method='OLS', criterion='MLSE')
    info = eng.describe()
    assert info['code_planes'] == 2
    # up to 4 092 sites the pass is the plain fp4 GEMM over the pre-expanded images (2 047 .. 4 092: the validity sum at 2^12 and the
    # congruence decode of dist_gemm.hip, round 6); longer alignments keep the bit-plane-fed matrix-core kernel
    assert info['fused_distance_pass'].startswith('fp4 gemm' if L <= 4092 else 'fp4 mfma'), info
    got = eng.place_sequences(qry)
    _compare(got, want, co, d, nodes, 'long alignment L=%d' % L)
    eng.close()
    if L <= 4092:  # ... and the same bytes from the bit-plane-fed kernel
        e2 = Engine(d.tree, ref, nodes, method='OLS', criterion='MLSE', debug=('no_dist_gemm',))
        assert e2.describe()['fused_distance_pass'].startswith('fp4 mfma')
        assert e2.place_sequences(qry).tobytes() == got.tobytes()
        e2.close()


def test_top_up_list_walked_in_slices(c2_full):
    """On the fused matrix-core path only the queries that need the top-up rule get full distance rows, and
    the row buffers hold a slice of the batch (an eighth, at least 2 048 rows): with a tiny threshold every
    one of 10 000 queries is on that list, which is then walked in five slices.  Same bytes as with
    batch-sized row buffers (APPLES_NO_SLIM_BATCH=1) and as the C oracle."""
    import subprocess
    d, nodes = c2_full
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from apples_amd import synth\n"
            "from apples_amd.engine import Engine\n"
            "d = synth.make_dataset(10000, 1000, 10000)\n"
            "nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)\n"
            "e = Engine(d.tree, d.ref_seqs, nodes, method='OLS', threshold=0.01, baseobs=30)\n"
            "sys.stdout.buffer.write(e.place_sequences(d.query_seqs).tobytes())\n" % ROOT)
    outs = []
    for env in ({}, {'APPLES_NO_SLIM_BATCH': '1'}):
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, env=dict(os.environ, **env), timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(r.stdout)
    assert len(outs[0]) == 10000 * 40 and outs[0] == outs[1]
    got = np.frombuffer(outs[0], dtype=np.dtype([('edge', '<i4'), ('flags', '<u4'), ('error', '<f8'), ('distal', '<f8'),
                                                 ('pendant', '<f8'), ('n_obs', '<i4'), ('n_valid', '<i4')], align=True))
    sample = np.arange(0, 10000, 200)
    co = COracle(d.tree, d.ref_seqs, nodes, method='OLS', threshold=0.01, baseobs=30, lut=jc69_lut(1000, 0.001),
                 threads=len(os.sched_getaffinity(0)))
    assert got[sample].tobytes() == co.place_sequences(d.query_seqs[sample]).tobytes()


def test_clade_blocks_with_a_pool_that_runs_dry_and_with_own_rows(monkeypatch):
    """The clustered route's clade blocks (whole subtrees of one cluster swept on a static schedule, sweep_lean.hip: k_blocks_up /
    k_blocks_down): a pool of 1 MB holds the tuples of a few tiles only -- the others' queries take those leaves one by one --,
    every query is a backbone leaf's own sequence once (its row is deleted, apples/PoolQueryWorker.py:63-66: the block that holds
    it is not whole), and a few queries equal a reference row (exact match inside a block).  Same bytes as without blocks."""
    from apples_amd import treecluster
    from apples_amd.fasta import Alignment
    from apples_amd.reference import ReducedReference
    d = synth.make_dataset(6000, 400, 700)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    ca = ReducedReference(Alignment(d.ref_names, d.ref_seqs), False, treecluster.grouped(d.tree, 0.24)).cluster_arrays()
    q = d.query_seqs.copy()
    self_rows = np.full(len(q), -1, np.int32)
    q[100:400] = d.ref_seqs[1000:1300]
    self_rows[100:400] = np.arange(1000, 1300)       # leave-one-out: the query is this reference row
    q[400:410] = d.ref_seqs[2000:2010]               # exact matches
    want = None
    for pool_mb, dbg in ((None, ('no_blocks',)), (None, ()), ('1', ())):
        if pool_mb:
            monkeypatch.setenv('APPLES_BLK_POOL_MB', pool_mb)
        eng = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='FM', debug=dbg)
        got = eng.place_sequences(q, self_rows)
        info = eng.describe()
        eng.close()
        assert (info['cluster_blocks'] > 0) == (dbg == ())
        if want is None:
            want = got
        else:
            assert got.tobytes() == want.tobytes(), (pool_mb, dbg)
    assert (want['flags'][400:410] & 1).all()  # F_EXACT


def test_knobs_are_per_context_not_per_process(c2_full):
    """apples_params.knobs: the library's tuning / test knobs are read once per context (over the process environment), not kept
    in function-local statics -- two contexts of one process differ in them (no hidden globals).  A context whose
    small sweep teams overflow early and whose lean pool runs dry beside a default one: different workspaces, the same bytes."""
    d, nodes = c2_full
    q = d.query_seqs[:2000]
    e1 = Engine(d.tree, d.ref_seqs, nodes, method='OLS', knobs={'SWEEP_CAP': 64, 'APPLES_LEAN_POOL_MB': 1, 'TOPUP_MIN_ROWS': 1})
    e2 = Engine(d.tree, d.ref_seqs, nodes, method='OLS')
    a, b = e1.place_sequences(q), e2.place_sequences(q)
    i1, i2 = e1.describe(), e2.describe()
    e1.close(); e2.close()
    assert a.tobytes() == b.tobytes()
    assert i1['sweep_team_cap'] != i2['sweep_team_cap'], (i1['sweep_team_cap'], i2['sweep_team_cap'])


def test_ragged_rows_of_the_clustered_route():
    """The clustered fused route's rows (csrc/common.h, Workspace::ragged): observation rows and rows of member distances sized for a
    query that observes every leaf are what bounds its device batch (20 B x leaves per query); from 65 536 reference rows on every query
    gets a row of 16 384 entries and only the queries whose flat member list is longer, or that leave the fast phases for the top-up
    rule / the general selection, take one of a few full-size rows.  80 000 leaves x L 200, self rows, exact matches, queries on the
    top-up rule: the same bytes with ragged rows (the default here), with small rows of 256 entries and a big row for every query (most take one),
    with a single big row (the block runs out and is placed once more with full rows, for good), without ragged rows; sampled against
    the C oracle.  A call of another entry point on the same context (full distance rows) reshapes the workspace and back."""
    import bench
    nq = 3000
    d = synth.make_dataset(80000, 200, nq)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    ca = bench.make_clusters(d, 0.2)
    q = d.query_seqs.copy()
    self_rows = np.full(nq, -1, np.int32)
    q[100:200] = d.ref_seqs[5000:5100]
    self_rows[100:200] = np.arange(5000, 5100)   # leave-one-out
    q[200:210] = d.ref_seqs[9000:9010]           # exact matches
    q[300:320, ::3] = ord('-')                   # fewer valid sites: more of them on the top-up rule
    kw = dict(clusters=ca, method='OLS', baseobs=40)
    eng = Engine(d.tree, d.ref_seqs, nodes, **kw)
    got = eng.place_sequences(q, self_rows)
    info = eng.describe()
    assert info['ragged_rows'] == 1 and info['row_small'] == 16384 and info['full_rows_for_good'] == 0 and info['cluster_blocks'] > 0, info
    batch_ragged = info['batch']
    _, dist = eng.distances(q[:4], want_counts=False)   # (another entry point: rows of n_slots distances)
    assert dist.shape[0] == 4 and dist.shape[1] >= 80000
    assert eng.place_sequences(q, self_rows).tobytes() == got.tobytes()
    eng.close()
    for knobs, expect in (({'APPLES_NO_RAGGED': 1}, dict(ragged_rows=0)), ({'APPLES_RAGGED_SMALL': 256, 'APPLES_RAGGED_BIG': 3100}, dict(ragged_rows=1, row_small=256, full_rows_for_good=0)),
                          ({'APPLES_RAGGED_SMALL': 256, 'APPLES_RAGGED_BIG': 1}, dict(ragged_rows=0, full_rows_for_good=1))):
        e = Engine(d.tree, d.ref_seqs, nodes, knobs=knobs, **kw)
        out = e.place_sequences(q, self_rows)
        i2 = e.describe()
        e.close()
        assert out.tobytes() == got.tobytes(), knobs
        for k, v in expect.items():
            assert i2[k] == v, (knobs, k, i2)
        if expect.get('row_small') == 256:
            assert i2['big_rows_last_batch'] > 100, i2
    assert batch_ragged >= nq
    sample = np.concatenate([np.arange(0, nq, 47), np.arange(100, 110), np.arange(200, 210), np.arange(300, 320)])
    co = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', baseobs=40, lut=jc69_lut(200, 0.001), threads=len(os.sched_getaffinity(0)))
    assert got[sample].tobytes() == co.place_sequences(q[sample], self_rows[sample]).tobytes()
