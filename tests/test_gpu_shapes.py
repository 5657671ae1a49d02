"""HIP path vs the C oracle at the FULL shapes AND query counts of BASELINE.json's configs 3, 4 and 5 (reference
size, alignment length, method; 100 000 / 50 000 queries, and for the -d config the 4 096-row block the bench holds
resident: 100 000 rows of 200 000 fp64 columns are 160 GB); a sample of at least 64 queries per config -- the ones
with the fewest and the most observed leaves plus a strided set -- is compared with the C oracle, and the whole pass
must not depend on how the device cuts it into batches.  Needs an MI355X (and some 10 GB of host memory for the
synthetic inputs)."""
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
NTHREADS = len(os.sched_getaffinity(0))


def _sample(got, n, extremes=8, strided=56):
    order = np.argsort(got['n_obs'], kind='stable')
    return np.unique(np.concatenate([order[:extremes], order[-extremes:],
                                     np.linspace(0, n - 1, strided).astype(np.int64)]))


def test_c3_shape_200k_leaves_100k_queries():
    """Config 3: 200 000-leaf backbone, L = 1000 nt, 100 000 queries, OLS/JC69, -f 0.2 -b 25: three device
    batches of 33 344 (what the bench times: `device_batches` / `batch_queries` of its line), so the sweep's pool and work
    lists are reused across batches, the
    top-up selection by segment minima runs on 200 k-slot rows and the host-buffer entry point streams its chunks.
    Checked: >= 64 sampled queries byte for byte against the C oracle; the resident and the streamed entry points
    agree; the result does not depend on the batch size."""
    nq = 100000
    d = synth.make_dataset(200000, 1000, nq)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS')
    info = eng.describe()
    assert info['n_nodes'] == 399999 and info['n_refs'] == 200000 and info['length'] == 1000
    # the fused pass of this workload is the GEMM form on the pre-expanded reference image (1 byte per site and slot)
    assert info['fused_distance_pass'] == 'fp4 gemm, linear threshold', info['fused_distance_pass']
    assert info['fp4_reference_image_bytes'] == 200192 * 16 * 64
    assert info['sweep_layout'] == 'lean'  # sweep_lean.hip: the default of big binary trees
    got = eng.place_sequences(d.query_seqs)          # host buffer in, host buffer out (streamed chunks)
    batch = eng.describe()['batch']
    assert batch < nq, 'expected at least two device batches, got batch = %d' % batch
    h, n = eng.upload_queries(d.query_seqs)          # resident block
    eng.place_resident(h)
    assert eng.fetch(h, n).tobytes() == got.tobytes()
    eng.free_queries(h)
    # rank 1's share of an 8-GPU job placed on its own: ONE small device batch (routing cut halved, 512-thread routed teams,
    # the top-up chain beside the sweep) against the same queries inside the full set's batches of 25 000
    shard = eng.place_sequences(d.query_seqs[12500:25000])
    assert shard.tobytes() == got[12500:25000].tobytes()
    h, n = eng.place_sequences_streamed(d.query_seqs[87500:])  # (what a rank of `bench.py --gpus 8` calls: placements left on the device)
    assert eng.fetch(h, n).tobytes() == got[87500:].tobytes()
    eng.free_queries(h)
    eng.close()
    # >= 1 024 queries byte for byte against the C oracle (round 3 compared 72 of the 100 000)
    sample = _sample(got, nq, extremes=32, strided=1000)
    assert len(sample) >= 1024
    co = COracle(d.tree, d.ref_seqs, nodes, method='OLS', lut=jc69_lut(1000, 0.001), threads=NTHREADS)
    want = co.place_sequences(d.query_seqs[sample])
    assert np.array_equal(got[sample]['edge'], want['edge'])
    assert got[sample].tobytes() == want.tobytes()
    assert ((got['flags'] & (F_EXACT | F_INSUFFICIENT)) != 0).sum() < nq // 50
    # another cut into device batches: same bytes
    e2 = Engine(d.tree, d.ref_seqs, nodes, method='OLS', max_batch=4096)
    again = e2.place_sequences(d.query_seqs[:9000])
    assert e2.describe()['batch'] == 4096
    e2.close()
    assert again.tobytes() == got[:9000].tobytes()
    # ALL 100 000 queries through a route that shares no distance or sweep code with the default: bit-plane-fed matrix-core
    # kernel instead of the GEMM form on the reference image, level loop with merged lists instead of sweep_lean.hip
    e3 = Engine(d.tree, d.ref_seqs, nodes, method='OLS', debug=('no_dist_gemm', 'no_sweep_lean'))
    i3 = e3.describe()
    assert i3['fused_distance_pass'] != info['fused_distance_pass'] and i3['sweep_layout'] != 'lean'
    other = e3.place_sequences(d.query_seqs)
    e3.close()
    assert other.tobytes() == got.tobytes()


def test_c4_shape_50k_leaves_L500_protein_fm():
    """Config 4: 50 000-leaf backbone, L = 500 aa, 50 000 queries, scoredist + FM;
    sampled queries against the C oracle: byte for byte (both sides sum the table values in fp64 left to
    right and take libm's log, csrc/libm_log.h; the reference's own summation order is BLAS-internal,
    SURVEY row a3, which is why the fixtures the reference wrote are tolerance-checked)."""
    nq = 50000
    d = synth.make_dataset(50000, 500, nq, protein=True)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    eng = Engine(d.tree, d.ref_seqs, nodes, protein=True, method='FM')
    assert eng.describe()['length'] == 500 and eng.describe()['n_refs'] == 50000
    got = eng.place_sequences(d.query_seqs)
    eng.close()
    e2 = Engine(d.tree, d.ref_seqs, nodes, protein=True, method='FM', max_batch=1024)
    again = e2.place_sequences(d.query_seqs[:3000])
    e2.close()
    assert again.tobytes() == got[:3000].tobytes()
    # every query once more through the route without the matrix-core filter (k_scoredist with the early exit, full rows for
    # the top-up list): same bytes
    e3 = Engine(d.tree, d.ref_seqs, nodes, protein=True, method='FM', debug=('no_sd_gemm',))
    other = e3.place_sequences(d.query_seqs)
    e3.close()
    assert other.tobytes() == got.tobytes()
    sample = _sample(got, nq, extremes=16, strided=500)
    assert len(sample) >= 512
    co = COracle(d.tree, d.ref_seqs, nodes, protein=True, method='FM', threads=NTHREADS)
    want = co.place_sequences(d.query_seqs[sample])
    g = got[sample]
    for f in ('edge', 'flags', 'n_obs', 'n_valid', 'error', 'distal', 'pendant'):
        assert np.array_equal(g[f], want[f]), f
    assert g.tobytes() == want.tobytes()  # (csrc/libm_log.h: the distances carry libm's log bits, as the oracle's do)


def test_c4_shape_clustered_default_protein_route():
    """Config 4's inputs through the command line's default route for -p (max-diameter clusters at 1.2 x -f, consensus
    representatives of the 21-symbol alphabet; apples/Reference.py:84-157): the fused route (distances to the representatives
    alone, cluster-major member distances) on all 50 000 queries; 256 + sampled queries against the C oracle (byte for byte),
    the first 6 000 also through full rows + general selection (round 4's route): same bytes."""
    from apples_amd import treecluster
    from apples_amd.fasta import Alignment
    from apples_amd.reference import ReducedReference
    nq = 50000
    d = synth.make_dataset(50000, 500, nq, protein=True)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    ca = ReducedReference(Alignment(d.ref_names, d.ref_seqs), True, treecluster.grouped(d.tree, 0.2 * 1.2)).cluster_arrays()
    eng = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, protein=True, method='FM')
    info = eng.describe()
    assert info['all_singleton'] == 0 and info['cluster_fused'] == 1 and 500 < info['n_reps'] < 5000, info
    got = eng.place_sequences(d.query_seqs)
    eng.close()
    e2 = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, protein=True, method='FM', debug=('no_fuse',))
    assert e2.describe()['cluster_fused'] == 0
    other = e2.place_sequences(d.query_seqs[:6000])
    e2.close()
    assert other.tobytes() == got[:6000].tobytes()
    sample = _sample(got, nq, extremes=16, strided=240)
    assert len(sample) >= 256
    co = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, protein=True, method='FM', threads=NTHREADS)
    want = co.place_sequences(d.query_seqs[sample])
    g = got[sample]
    for f in ('edge', 'flags', 'n_obs', 'n_valid', 'error', 'distal', 'pendant'):
        assert np.array_equal(g[f], want[f]), f
    assert g.tobytes() == want.tobytes()  # (csrc/libm_log.h: the distances carry libm's log bits, as the oracle's do)


def test_c3_shape_other_criteria_and_negative_branches():
    """Config 3's backbone with -c ME / HYBRID and with -n (apples/Algorithm.py:76-101, apples/util.py:32-50): 4 096 queries
    placed, a sample of 256 + byte for byte against the C oracle.  All of them ride the lean sweep's kernels (HYBRID: every edge's
    solution kept in the entries, ranked after the top-down pass); HYBRID is crossed, all 4 096 queries byte for byte, with the
    level loop over per-edge records (sweep.hip, the `hybrid_records` switch: the form of rounds 1 - 4), for OLS and for FM."""
    nq = 4096
    d = synth.make_dataset(200000, 1000, nq)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    lut = jc69_lut(1000, 0.001)
    for criterion, negative in (('ME', False), ('HYBRID', False), ('MLSE', True), ('HYBRID', True)):
        eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS', criterion=criterion, negative=negative)
        got = eng.place_sequences(d.query_seqs)
        layout = eng.describe()['sweep_layout']
        eng.close()
        assert layout == 'lean', (criterion, layout)
        sample = _sample(got, nq, extremes=16, strided=240)
        assert len(sample) >= 256
        co = COracle(d.tree, d.ref_seqs, nodes, method='OLS', criterion=criterion, negative=negative, lut=lut, threads=NTHREADS)
        want = co.place_sequences(d.query_seqs[sample])
        assert np.array_equal(got[sample]['edge'], want['edge']), (criterion, negative)
        assert got[sample].tobytes() == want.tobytes(), (criterion, negative)
        if criterion == 'HYBRID':
            for method in ('OLS', 'FM') if not negative else ('OLS',):
                e1 = Engine(d.tree, d.ref_seqs, nodes, method=method, criterion='HYBRID', negative=negative)
                a = e1.place_sequences(d.query_seqs)
                e1.close()
                e2 = Engine(d.tree, d.ref_seqs, nodes, method=method, criterion='HYBRID', negative=negative, debug=('hybrid_records',))
                b = e2.place_sequences(d.query_seqs)
                assert e2.describe()['sweep_layout'] != 'lean'
                e2.close()
                assert a.tobytes() == b.tobytes(), (method, negative, int((a['edge'] != b['edge']).sum()))
                if method == 'OLS':
                    assert a.tobytes() == got.tobytes()


def test_c5_shape_200k_column_distance_table_bme():
    """Config 5: -d input, 200 000 columns, BME.  The bench's block: 4 096 table rows (6.5 GB) resident, rows with
    missing values, an exact hit, a row with nothing observed; 256 rows (the special ones, the extremes and a strided
    set) byte for byte against the C oracle; the host-buffer entry point in several device batches (max_batch = 96)
    on the first 300 rows and with the default batch on the first 100: identical bytes."""
    nq = 4096
    d = synth.make_dataset(200000, 8, nq)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    ix = synth.TreeIndex(d.tree)
    D = synth.fast_distance_rows(d.tree, ix, d.query_leaf, d.query_pendant, list(range(nq)))
    D[5, ::3] = -1.0
    D[6, :] = -1.0
    D[7, 123456] = 0.0
    eng = Engine(d.tree, None, method='BME')
    h, n = eng.upload_table(D, nodes)
    eng.place_resident(h)
    got = eng.fetch(h, n)
    eng.free_queries(h)
    eng.close()
    e1 = Engine(d.tree, None, method='BME', max_batch=96)
    assert e1.place_distances(D[:300], nodes).tobytes() == got[:300].tobytes()
    assert e1.describe()['batch'] == 96
    e1.close()
    e2 = Engine(d.tree, None, method='BME')
    assert e2.place_distances(D[:100], nodes).tobytes() == got[:100].tobytes()
    e2.close()
    sample = np.unique(np.concatenate([np.arange(16), _sample(got, nq, extremes=16, strided=208)]))
    assert len(sample) >= 200
    want = COracle(d.tree, method='BME', threads=NTHREADS).place_distances(np.ascontiguousarray(D[sample]), nodes)
    assert got[sample].tobytes() == want.tobytes()
    assert got[7]['flags'] & F_EXACT and got[6]['flags'] & F_INSUFFICIENT


def test_c5_one_shard_of_eight_12500_rows_resident():
    """Config 5 as one rank of the 8-GPU job sees it: 12 500 of the 100 000 table rows (20 GB of fp64) resident, placed as a
    pipeline of sub-batches (selection of one beside the sweep of the one before, two sets of batch buffers).  512 rows -- the
    special ones, the extremes, a strided set -- byte for byte against the C oracle; the first 3 000 rows through the host-buffer
    entry point (one stream, other batch cuts): identical bytes."""
    nq = 12500
    d = synth.make_dataset(200000, 8, nq)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    ix = synth.TreeIndex(d.tree)
    D = np.empty((nq, 200000))
    for lo in range(0, nq, 2500):  # (block by block: the generator's temporaries stay small)
        D[lo:lo + 2500] = synth.fast_distance_rows(d.tree, ix, d.query_leaf, d.query_pendant, list(range(lo, lo + 2500)), seed_noise=7 + lo)
    D[5, ::3] = -1.0
    D[6, :] = -1.0
    D[7, 123456] = 0.0
    D[9000, 1::2] = -1.0
    eng = Engine(d.tree, None, method='BME')
    h, n = eng.upload_table(D, nodes)
    eng.place_resident(h)
    got = eng.fetch(h, n)
    eng.place_resident(h)                       # a second pass over the same resident block: same bytes
    assert eng.fetch(h, n).tobytes() == got.tobytes()
    eng.free_queries(h)
    eng.close()
    e1 = Engine(d.tree, None, method='BME', max_batch=1024)
    assert e1.place_distances(D[:3000], nodes).tobytes() == got[:3000].tobytes()
    e1.close()
    sample = np.unique(np.concatenate([np.arange(16), [9000], _sample(got, nq, extremes=24, strided=460)]))
    assert len(sample) >= 480
    want = COracle(d.tree, method='BME', threads=NTHREADS).place_distances(np.ascontiguousarray(D[sample]), nodes)
    assert got[sample].tobytes() == want.tobytes()
    assert got[7]['flags'] & F_EXACT and got[6]['flags'] & F_INSUFFICIENT


def test_resident_table_is_refused_after_the_column_layout_changed():
    """A resident -d block was permuted with the column order of its upload; placing another table
    with different columns of the same number replaces that order.  The old block must be refused,
    not silently run against the wrong node map."""
    d = synth.make_dataset(400, 8, 8)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    ix = synth.TreeIndex(d.tree)
    D = synth.fast_distance_rows(d.tree, ix, d.query_leaf, d.query_pendant, list(range(8)))
    eng = Engine(d.tree, None, method='BME')
    h, n = eng.upload_table(D, nodes)
    eng.place_resident(h)
    first = eng.fetch(h, n)
    perm = np.random.default_rng(1).permutation(len(nodes))
    other = eng.place_distances(D[:, perm], nodes[perm])      # same table, columns in another order
    assert other.tobytes() == first.tobytes()
    with pytest.raises(RuntimeError, match='column layout changed'):
        eng.place_resident(h)
    eng.free_queries(h)
    h2, n = eng.upload_table(D, nodes)
    eng.place_resident(h2)
    assert eng.fetch(h2, n).tobytes() == first.tobytes()
    eng.close()


def test_query_matrix_of_another_length_is_refused():
    """The reference fails loudly when query and reference lengths differ (numpy elementwise compare,
    apples/distance.py:733); bytes must never be re-chunked into a different number of queries."""
    d = synth.make_dataset(300, 64, 4)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS')
    bad = np.full((4, 32), ord('A'), np.uint8)   # 4 x 32 bytes would re-chunk into 2 x 64
    for call in (eng.place_sequences, eng.upload_queries, eng.distances, eng.place_sequences_streamed):
        with pytest.raises(ValueError, match='reference alignment length'):
            call(bad)
    assert len(eng.place_sequences(d.query_seqs)) == 4
    eng.close()


def test_exotic_symbols_in_a_later_chunk_of_a_streamed_block():
    """apples_place_from_sequences streams the caller's buffer chunk by chunk; a symbol beyond ACGT-
    in a later chunk is only seen after earlier chunks ran on the 2-plane images, which took it for a gap.  The
    call must then place those queries once more through the 8-plane kernels ("any other byte is an ordinary
    symbol", apples/distance.py:733) -- and since round 6 the context keeps its matrix-core forms (one `*` used to
    re-pack it to 8 planes for good)."""
    d = synth.make_dataset(1500, 200, 400)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    q = d.query_seqs.copy()
    q[333, 17] = ord('*')
    q[334, :50] = ord('N')
    eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS', max_batch=64)
    assert eng.describe()['code_planes'] == 2
    got = eng.place_sequences(q)
    info = eng.describe()
    assert info['code_planes'] == 2 and info['eight_plane_copy'] == 1 and info['fused_distance_pass'].startswith('fp4'), info
    co = COracle(d.tree, d.ref_seqs, nodes, method='OLS', lut=jc69_lut(200, 0.001), threads=NTHREADS)
    want = co.place_sequences(q)
    assert got.tobytes() == want.tobytes()
    # the same block resident (the flags are known before the pass), and a block where most queries carry such bytes (the whole
    # block through the 8-plane kernels)
    h, n = eng.upload_queries(q)
    eng.place_resident(h)
    assert eng.fetch(h, n).tobytes() == want.tobytes()
    eng.free_queries(h)
    q2 = d.query_seqs.copy()
    q2[::2, 5] = ord('.')
    q2[1::4, 190:] = ord('?')
    got2 = eng.place_sequences(q2)
    assert got2.tobytes() == co.place_sequences(q2).tobytes()
    assert eng.place_sequences(q).tobytes() == want.tobytes()  # (and the fast route again afterwards)
    eng.close()


def test_exotic_symbols_in_the_reference_rows_stay_on_the_matrix_cores():
    """Reference rows with bytes beyond ACGT- (`.` as a gap character is ordinary in SILVA / ARB exports; apples/distance.py:733-737
    compares them as bytes): the fused pass keeps running on the matrix cores with such bytes as gaps, k_exotic_fix gives the
    survivors on those rows their exact counts, full rows (top-up, apples_distances) come from the 8-plane copy.  Sparse (one `.`
    per 1 000 sites in 5 % of the rows: bench.py's c3-dots), dense (every row, runs of `.` at both ends), and under -V 0.5, where the
    smaller valid count of the gap form could have kept a pair out; queries with such bytes among them.  Byte for byte against the
    C oracle, and equal to the route without the matrix-core pass."""
    rng = np.random.default_rng(21)
    for n, L, nq, V, dense in ((3000, 1000, 700, 0.001, False), (1500, 300, 500, 0.001, True), (1200, 400, 300, 0.5, True), (40000, 500, 600, 0.001, False)):
        d = synth.make_dataset(n, L, nq, gap_rate=0.3 if V > 0.1 else 0.05)
        nodes = np.array([d.tree.name_to_node[x] for x in d.ref_names], np.int32)
        ref = d.ref_seqs.copy()
        if dense:
            for i in range(n):
                a, b = rng.integers(0, L // 6, size=2)
                ref[i, :a] = ord('.'); ref[i, L - b:] = ord('.')
                ref[i, rng.integers(0, L, size=3)] = ord('?')
        else:
            rows = np.nonzero(rng.random(n) < 0.05)[0]
            ref[rows, rng.integers(0, L, size=len(rows))] = ord('.')
        q = d.query_seqs.copy()
        q[0, 7] = ord('?')
        q[5, :40] = ord('.')
        q[6] = ref[11]          # an exact match of a row with such bytes
        eng = Engine(d.tree, ref, nodes, method='OLS', overlap=V)
        info = eng.describe()
        assert info['code_planes'] == 2 and info['eight_plane_copy'] == 1 and info['fused_distance_pass'].startswith('fp4'), info
        got = eng.place_sequences(q)
        counts, dist = eng.distances(q[:16])
        eng.close()
        co = COracle(d.tree, ref, nodes, method='OLS', overlap=V, lut=jc69_lut(L, V), threads=NTHREADS)
        want = co.place_sequences(q)
        assert got.tobytes() == want.tobytes(), (n, L, V, dense, int((got['edge'] != want['edge']).sum()))
        assert dist.tobytes() == co.distances(q[:16])[:, :n].tobytes()
        e2 = Engine(d.tree, ref, nodes, method='OLS', overlap=V, debug=('no_fuse',))
        assert e2.place_sequences(q).tobytes() == got.tobytes()
        e2.close()


def test_scan_sweep_at_c3_shape_and_on_a_deep_tree(monkeypatch):
    """The scan formulation of the sweep (the sweep_scan switch: id-sorted leaves, Euler-tour lowest common
    ancestors, prefix counts instead of a node map) returns the level loop's bytes at the 200 000-leaf
    shape; a tree deeper than its 254-level tables (a caterpillar) quietly keeps the level loop."""
    from apples_amd.tree import parse_newick
    d = synth.make_dataset(200000, 1000, 3000)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS')
    assert eng.describe()['sweep'] == 'levels'
    want = eng.place_sequences(d.query_seqs)
    eng.close()
    eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS', debug=('sweep_scan',))
    assert eng.describe()['sweep'] == 'scan'
    got = eng.place_sequences(d.query_seqs)
    eng.close()
    assert got.tobytes() == want.tobytes()
    n = 400
    rng = np.random.default_rng(9)
    s = 't0:0.01'
    for i in range(1, n):
        s = '(%s,t%d:%.4f):%.4f' % (s, i, rng.uniform(0.005, 0.02), rng.uniform(0.001, 0.01))
    tree = parse_newick(s[:s.rindex(':')] + ';')
    assert int(np.max(tree.level)) > 254
    base = rng.integers(0, 4, size=300)
    alpha = np.frombuffer(b'ACGT', np.uint8)
    names = ['t%d' % i for i in range(n)]
    seqs = np.empty((n, 300), np.uint8)
    for i in range(n):
        row = base.copy()
        hit = rng.random(300) < 0.06
        row[hit] = rng.integers(0, 4, size=int(hit.sum()))
        seqs[i] = alpha[row]
    qry = seqs[rng.integers(0, n, size=40)].copy()
    qry[rng.random(qry.shape) < 0.03] = ord('G')
    nodes = np.array([tree.name_to_node[x] for x in names], np.int32)
    eng = Engine(tree, seqs, nodes, method='FM', threshold=0.08, baseobs=10)
    assert eng.describe()['sweep'] == 'levels'
    got = eng.place_sequences(qry)
    eng.close()
    want = COracle(tree, seqs, nodes, method='FM', threshold=0.08, baseobs=10, lut=jc69_lut(300, 0.001),
                   threads=NTHREADS).place_sequences(qry)
    assert got.tobytes() == want.tobytes()


def _clustered(d, thr):
    from apples_amd import treecluster
    from apples_amd.fasta import Alignment
    from apples_amd.reference import ReducedReference
    return ReducedReference(Alignment(d.ref_names, d.ref_seqs), False, treecluster.grouped(d.tree, thr * 1.2)).cluster_arrays()


@pytest.mark.parametrize('thr,b', [(0.2, 25), (0.02, 25), (0.05, 400)])
def test_clustered_route_fused_by_representatives_against_c_oracle(thr, b, monkeypatch):
    """The command line's default route (max-diameter clusters, consensus representatives, heap-ordered
    cluster expansion, apples/Reference.py:117-157) on its fused path: matrix-core pass over the
    representatives, members of the accepted clusters expanded per query, slow list for queries whose
    accepted clusters hold fewer than -b valid distances (small thresholds send most queries there).
    Byte for byte against the C oracle, and against the unfused path (full rows + general selection)."""
    d = synth.make_dataset(6000, 700, 700)
    q = d.query_seqs.copy()
    q[5] = d.ref_seqs[77]            # exact hit through a cluster member
    q[6] = ord('-')                  # nothing observed
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    ca = _clustered(d, thr)
    assert len(ca[0]) > 20
    want = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', threshold=thr, baseobs=b, lut=jc69_lut(700, 0.001),
                   threads=NTHREADS).place_sequences(q)
    self_rows = np.full(len(q), -1, np.int32)
    eng = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', threshold=thr, baseobs=b)
    got = eng.place_sequences(q)
    eng.close()
    assert got.tobytes() == want.tobytes()
    eng = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', threshold=thr, baseobs=b, debug=('no_fuse',))
    unfused = eng.place_sequences(q, self_rows)
    eng.close()
    assert unfused.tobytes() == want.tobytes()
    assert got[5]['flags'] & F_EXACT and got[6]['flags'] & F_INSUFFICIENT


def test_clustered_route_at_c3_shape(monkeypatch):
    """The same at 200 000 leaves (about 5 000 clusters of 40): fused and unfused paths agree on 3 000
    queries, a sample agrees with the C oracle."""
    d = synth.make_dataset(200000, 1000, 3000)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    ca = _clustered(d, 0.2)
    eng = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS')
    got = eng.place_sequences(d.query_seqs)
    eng.close()
    eng = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', debug=('no_fuse',))
    unfused = eng.place_sequences(d.query_seqs[:600])
    eng.close()
    assert unfused.tobytes() == got[:600].tobytes()
    sample = _sample(got, 3000, extremes=4, strided=24)
    want = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', lut=jc69_lut(1000, 0.001),
                   threads=NTHREADS).place_sequences(d.query_seqs[sample])
    assert got[sample].tobytes() == want.tobytes()


@pytest.mark.parametrize('thr,b', [(0.2, 25), (0.03, 60)])
def test_scoredist_fused_threshold_compaction_equals_full_rows(thr, b):
    """scoredist with singleton clusters keeps, like JC69, only the entries inside the threshold in the
    distance kernel's epilogue; queries that need the top-up rule get full rows (listed mode).  Same bytes as
    the unfused route (the no_fuse switch: full rows + general selection); edges and counts equal to the C
    oracle's, lengths within 1e-9."""
    d = synth.make_dataset(5000, 300, 900, protein=True)
    q = d.query_seqs.copy()
    q[3] = d.ref_seqs[11]
    q[4] = ord('-')
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    eng = Engine(d.tree, d.ref_seqs, nodes, protein=True, method='FM', threshold=thr, baseobs=b)
    got = eng.place_sequences(q)
    eng.close()
    e2 = Engine(d.tree, d.ref_seqs, nodes, protein=True, method='FM', threshold=thr, baseobs=b, debug=('no_fuse',))
    unfused = e2.place_sequences(q)
    e2.close()
    assert unfused.tobytes() == got.tobytes()
    want = COracle(d.tree, d.ref_seqs, nodes, protein=True, method='FM', threshold=thr, baseobs=b, threads=NTHREADS).place_sequences(q)
    for f in ('edge', 'flags', 'n_obs', 'n_valid'):
        assert np.array_equal(got[f], want[f]), f
    for f in ('error', 'distal', 'pendant'):
        np.testing.assert_allclose(got[f], want[f], rtol=1e-9, atol=1e-15, err_msg=f)
    assert got[3]['flags'] & F_EXACT and got[4]['flags'] & F_INSUFFICIENT


def test_empty_query_block():
    d = synth.make_dataset(300, 64, 4)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS')
    none = np.zeros((0, 64), np.uint8)
    assert len(eng.place_sequences(none)) == 0
    h, n = eng.place_sequences_streamed(none)
    assert n == 0 and len(eng.fetch(h, 0)) == 0
    eng.free_queries(h)
    assert len(eng.place_sequences(d.query_seqs)) == 4
    eng.close()


@pytest.mark.parametrize('shape', ['unrooted', 'polytomies', 'deep'])
def test_c3_backbone_in_the_shapes_real_trees_have(shape):
    """Config 3's backbone as real inputs come (bench.py: variant_dataset): unrooted (a root trifurcation: what FastTree prints,
    what both of the reference's example backbones have), 1 % of the internal nodes dissolved into polytomies
    (apples/OLS.py:36,59 loop over any number of children), hung on a caterpillar spine of 400 (427 levels).  Until round 6
    each of them fell off the lean sweep (and the clade blocks) onto the level loop.  Checked on 8 192 queries, singleton
    clusters and the command line's default route: the route taken (`sweep_layout`, `cluster_blocks`), the level loop's bytes
    (`no_sweep_lean`; clustered: `no_blocks`), 64 + sampled queries byte for byte against the C oracle."""
    import bench
    nq = 8192
    base = synth.make_dataset(200000, 1000, nq, spine=400 if shape == 'deep' else 0)
    d = base if shape == 'deep' else bench.variant_dataset(base, shape, 200000, 1000, nq, False)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    eng = Engine(d.tree, d.ref_seqs, nodes, method='OLS')
    info = eng.describe()
    assert info['sweep_layout'] == 'lean', info
    assert (info['max_children'] > 2) == (shape != 'deep') and (info['height'] > 254) == (shape == 'deep'), info
    got = eng.place_sequences(d.query_seqs)
    eng.close()
    e2 = Engine(d.tree, d.ref_seqs, nodes, method='OLS', debug=('no_sweep_lean',))
    assert e2.describe()['sweep_layout'] != 'lean'
    other = e2.place_sequences(d.query_seqs)
    e2.close()
    assert other.tobytes() == got.tobytes()
    sample = _sample(got, nq, extremes=8, strided=56)
    co = COracle(d.tree, d.ref_seqs, nodes, method='OLS', lut=jc69_lut(1000, 0.001), threads=NTHREADS)
    assert co.place_sequences(d.query_seqs[sample]).tobytes() == got[sample].tobytes()
    # the command line's default route: clusters + consensus representatives, clade blocks
    ca = bench.make_clusters(d, 0.2)
    e3 = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS')
    i3 = e3.describe()
    assert i3['sweep_layout'] == 'lean' and i3['cluster_fused'] == 1 and i3['cluster_blocks'] > 1000, i3
    gc = e3.place_sequences(d.query_seqs)
    e3.close()
    e4 = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', debug=('no_blocks', 'no_sweep_lean'))
    oc = e4.place_sequences(d.query_seqs)
    e4.close()
    assert oc.tobytes() == gc.tobytes()
    sample = _sample(gc, nq, extremes=8, strided=56)
    cc = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', lut=jc69_lut(1000, 0.001), threads=NTHREADS)
    assert cc.place_sequences(d.query_seqs[sample]).tobytes() == gc[sample].tobytes()


@pytest.mark.parametrize('method', ['OLS', 'FM', 'BME', 'BE'])
def test_unrooted_backbones_every_method_and_criterion(method):
    """An unrooted binary backbone -- a root with three children, the tree's only polytomy: what every tree-building program prints.
    The lean sweep's kernels for trees with polytomies against the level loop (`no_sweep_lean`) for every method, MLSE / ME / HYBRID,
    with and without `-n`, singleton and clustered, with root children that are leaves (a three-leaf star, a tiny tree) among the
    shapes; sampled against the C oracle.  (Round 6 also tried the binary tree's kernels with a step of their own for the root: the
    same bytes, slower on both kinds of tree -- profiles/r06_root3_ab.txt -- and not kept.)"""
    import bench
    for n, nq in ((3, 40), (7, 60), (12000, 3000)):
        d = synth.make_dataset(n, 300, nq, seed_tree=77 + n)
        d.tree = synth.reshape_tree(d.tree, 'unrooted')
        nodes = np.array([d.tree.name_to_node[x] for x in d.ref_names], np.int32)
        ca = bench.make_clusters(d, 0.2) if n > 100 else None
        for crit, neg in (('MLSE', False), ('ME', True), ('HYBRID', False)):
            for clusters in ((None, ca) if ca is not None else (None,)):
                kw = dict(clusters=clusters, method=method, criterion=crit, negative=neg, threshold=0.3)
                e = Engine(d.tree, d.ref_seqs, nodes, **kw)
                info = e.describe()
                got = e.place_sequences(d.query_seqs)
                e.close()
                assert info['max_children'] == 3, info
                e2 = Engine(d.tree, d.ref_seqs, nodes, debug=('no_sweep_lean', 'no_blocks'), **kw)
                want = e2.place_sequences(d.query_seqs)
                e2.close()
                assert got.tobytes() == want.tobytes(), (n, crit, neg, clusters is not None)
                if crit != 'HYBRID':
                    sample = np.arange(0, nq, 7)
                    co = COracle(d.tree, d.ref_seqs, nodes, clusters=clusters, method=method, criterion=crit, negative=neg, threshold=0.3,
                                 lut=jc69_lut(300, 0.001), threads=NTHREADS)
                    assert co.place_sequences(d.query_seqs[sample]).tobytes() == got[sample].tobytes(), (n, crit, neg, clusters is not None)


def test_clustered_route_beyond_229376_slots():
    """The command line's default route on a reference of more than 229 376 rows (k_select_clusters' bitmap over the slots held that
    many until round 6; 524 288 now: runs of 32 words per thread): 270 000 leaves x L 300, clusters + consensus representatives,
    clade blocks.  The fused route's bytes against full rows + general selection on 3 000 queries, 64 + sampled ones against the C
    oracle."""
    import bench
    nq = 3000
    d = synth.make_dataset(270000, 300, nq)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    ca = bench.make_clusters(d, 0.2)
    eng = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS')
    info = eng.describe()
    assert info['n_refs'] == 270000 and info['cluster_fused'] == 1 and info['cluster_blocks'] > 1000 and info['sweep_layout'] == 'lean', info
    got = eng.place_sequences(d.query_seqs)
    eng.close()
    e2 = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', debug=('no_fuse',))
    assert e2.describe()['cluster_fused'] == 0
    assert e2.place_sequences(d.query_seqs[:1000]).tobytes() == got[:1000].tobytes()
    e2.close()
    # more than 5 120 clusters: the queries that accept more than 512 of them go through the third form of the selection's phases
    # (k_select_clusters<.., HUGE_CAP>: lists in global scratch) and keep their clade blocks; without it (the knob) they took full
    # rows and the general selection -- the same bytes either way
    assert info['n_reps'] > 5120, info
    e3 = Engine(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', knobs={'APPLES_NO_CLUSTER_HUGE': 1})
    assert e3.place_sequences(d.query_seqs).tobytes() == got.tobytes()
    e3.close()
    assert int(np.max(got['n_obs'])) > 20000  # (such queries are there)
    sample = _sample(got, nq, extremes=8, strided=56)
    cc = COracle(d.tree, d.ref_seqs, nodes, clusters=ca, method='OLS', lut=jc69_lut(300, 0.001), threads=NTHREADS)
    assert cc.place_sequences(d.query_seqs[sample]).tobytes() == got[sample].tobytes()
