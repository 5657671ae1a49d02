"""CPU-only tests of the host side that mirrors run_apples.py: options, distance-table reader,
jplace join, consensus representatives, query sharding."""
import io
import os
import sys

import numpy as np
import pytest

from helpers import DATA, GOLD, ROOT, load_json

sys.path.insert(0, ROOT)
from apples_amd.options import options_config  # noqa: E402
from apples_amd.jplace import join_jplace, finish, dumps  # noqa: E402
from apples_amd.reference import ReducedReference, consensus, read_treecluster  # noqa: E402
from apples_amd.fasta import read_alignment  # noqa: E402
from apples_amd.worker import _shards  # noqa: E402
import run_apples  # noqa: E402


def test_option_defaults_match_reference():
    o, _ = options_config(['-t', 'x.nwk', '-s', 'r.fa', '-q', 'q.fa'])
    assert o.method_name == 'FM' and o.criterion_name == 'MLSE'          # OptionsRun.py:32-47
    assert o.filt_threshold == 0.2 and o.base_observation_threshold == 25  # OptionsBasic.py:44, OptionsRun.py:59
    assert o.minimum_alignment_overlap == 0.001 and not o.negative_branch and not o.exclude_intplace
    assert o.reestimate_backbone and o.num_thread >= 1


def test_option_validation_matches_reference():
    with pytest.raises(ValueError):
        options_config(['-t', 'x', '-d', 'd.mat', '-s', 'r.fa'])     # OptionsRun.py:90-91
    with pytest.raises(ValueError):
        options_config(['-s', 'r.fa', '-q', 'q.fa'])                 # OptionsRun.py:106-107
    with pytest.raises(ValueError):
        options_config(['-t', 'x', '-s', 'r', '-q', 'q.fa', '-x', 'e.fa'])  # OptionsRun.py:108-109
    o, _ = options_config(['-t', 'x', '-d', 'd.mat'])
    assert o.reestimate_backbone is False                              # OptionsRun.py:88-89


def test_read_dismat_small_and_data():
    with open(os.path.join(DATA, 'small_dist.mat')) as f:
        names, cols, D = run_apples.read_dismat(f)
    assert names == ['myquery'] and cols == ['A', 'B', 'C', 'D', 'E']
    assert D.tolist() == [[-1.0, -1.0, 0.2, 0.7, 0.7]]
    with open(os.path.join(DATA, 'dist.mat')) as f:
        names, cols, D = run_apples.read_dismat(f)
    assert len(names) == 10 and len(cols) == 500 and D.shape == (10, 500)
    # dict(zip(tags, values)) semantics: a repeated column keeps its first position and last value
    names, cols, D = run_apples.read_dismat(io.StringIO('\tA B A\nq 1 2 3\n'))
    assert cols == ['A', 'B'] and D.tolist() == [[3.0, 2.0]]


def _res(name, edge):
    return {'placements': [{'p': [[edge, 0, 1, 0, 0]], 'n': [name]}]}


def test_join_jplace_keep_first_quirk():
    # apples/jutil.py:11-18: unplaceable results vanish, except the first when there are several
    j = join_jplace([_res('a', -1), _res('b', 3), _res('c', -1), _res('d', 5)])
    assert [p['n'][0] for p in j['placements']] == ['a', 'b', 'd']
    assert join_jplace([_res('a', -1)])['placements'] == []
    assert len(join_jplace([_res('a', 2)])['placements']) == 1
    out = dumps(finish(join_jplace([_res('a', 2)]), '(A,B);', ['run_apples.py', '-t', 'x']))
    assert out.endswith('\n') and '"version": 3' in out
    import json
    assert list(json.loads(out)) == ['fields', 'metadata', 'placements', 'tree', 'version']


def test_consensus_matches_reference_fixture():
    ref = read_alignment(os.path.join(DATA, 'ref.fa'), False, False)
    reps = load_json('g2_selection.json')['clade_clusters']
    n = 0
    for r in reps:
        if len(r['members']) > 1:
            rows = [ref.index[m] for m in r['members']]
            assert consensus(ref.seqs[rows], False).tobytes().decode() == r['cons']
            n += 1
    assert n > 5
    # ties go to the first symbol in A C G T - order; symbols outside the alphabet are not counted
    arr = np.frombuffer(b'AC*' b'CA*', np.uint8).reshape(2, 3)
    assert consensus(arr, False).tobytes() == b'AAA'


def test_protein_consensus_matches_reference_fixture(tmp_path):
    """The 21-symbol alphabet of apples/PoolRepresentativeWorker.py:33-58 (A C D E ... Y -: alphabetical, not a2i's
    order) against rows the reference's own _find_representative(prot_flag=True) produced (g10), numpy and native."""
    import prot_cases
    from apples_amd.reference import _consensus_rows
    ref_fp, _, _ = prot_cases.write_case(str(tmp_path))
    ref = read_alignment(ref_fp, True, False)
    reps = [r for r in load_json('g10_prot_clustered.json')['clade_clusters'] if len(r['members']) > 1]
    assert len(reps) > 50
    mrow, groups = [], []
    for r in reps:
        rows = [ref.index[m] for m in r['members']]
        assert consensus(ref.seqs[rows], True).tobytes().decode() == r['cons']
        groups.append((len(mrow), len(mrow) + len(rows)))
        mrow += rows
    native = _consensus_rows(ref.seqs, np.array(mrow, np.int32), groups, True)
    assert [bytes(x).decode() for x in native] == [r['cons'] for r in reps]
    assert any('-' in r['cons'] for r in reps)  # a gap can win a column
    # ties go to the first symbol in A C D E F G H I K L M N P Q R S T V W Y - order; other symbols are not counted
    arr = np.frombuffer(b'RC*x' b'CR*x', np.uint8).reshape(2, 4)
    assert consensus(arr, True).tobytes() == b'CCAA'


def test_reduced_reference_from_treecluster_file(tmp_path):
    ref = read_alignment(os.path.join(DATA, 'ref.fa'), False, False)
    names = ref.names
    p = tmp_path / 'tc.txt'
    with open(p, 'w') as f:
        f.write('SequenceName\tClusterNumber\n')
        for i, n in enumerate(names):
            f.write('%s\t%s\n' % (n, '-1' if i >= 6 else ('10' if i < 3 else '2')))
    cl = read_treecluster(str(p))
    assert [k for k, _ in cl] == ['-1', '10', '2']  # sorted as strings (Reference.py:97)
    rr = ReducedReference(ref, False, cl)
    cons, rep_row, moff, mrow = rr.cluster_arrays()
    assert len(cons) == 2 and len(rep_row) == len(names) - 6 + 2 and moff[-1] == len(names)
    assert list(rep_row[-2:]) == [len(names), len(names) + 1] and list(mrow[-6:]) == [0, 1, 2, 3, 4, 5]


def test_shards_cover_in_order():
    for n in (0, 1, 7, 100000):
        for parts in (1, 2, 8):
            sh = _shards(n, parts)
            assert sh[0][0] == 0 and sh[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
            assert max(h - l for l, h in sh) - min(h - l for l, h in sh) <= 1


def test_max_diameter_clustering_properties():
    """apples_amd/treecluster.py (TreeCluster 'max' method restated; parity unpinned): every leaf in
    exactly one cluster, pairwise distances inside a cluster <= t, singletons labelled -1, grouping
    as apples/Reference.py:93-100."""
    import itertools
    from apples_amd import treecluster as tc, synth
    from apples_amd.tree import read_tree
    for path, t in ((os.path.join(DATA, 'backbone.nwk'), 0.24), (os.path.join(DATA, 'prot', 'backbone.nwk'), 0.72)):
        tree = read_tree(path)
        clusters = tc.max_clusters(tree, t)
        names = [n for c in clusters for n in c]
        assert sorted(names) == sorted(tree.labels[v] for v in tree.leaves)
        rd = synth.TreeIndex(tree).rd

        def dist(a, b):
            anc = set()
            u = a
            while u >= 0:
                anc.add(u)
                u = tree.parent[u]
            u = b
            while u not in anc:
                u = tree.parent[u]
            return rd[a] + rd[b] - 2 * rd[u]
        if (tree.edge_len >= 0).all():  # the diameter guarantee presumes non-negative branch lengths
            for c in clusters:
                ids = [tree.name_to_node[n] for n in c][:25]
                for a, b in itertools.combinations(ids, 2):
                    assert dist(a, b) <= t + 1e-12
        g = tc.grouped(tree, t)
        assert [k for k, _ in g] == sorted(k for k, _ in g)  # ids sorted as strings
        assert sum(len(m) for _, m in g) == tree.n_leaves
        if any(len(c) == 1 for c in clusters):
            assert g[0][0] == '-1'
    # a huge threshold puts everything in one cluster; zero makes every leaf a singleton
    tree = read_tree(os.path.join(DATA, 'small_backbone.nwk'))
    assert tc.grouped(tree, 100.0) == [('1', ['A', 'B', 'C', 'D', 'E'])]
    g0 = tc.grouped(tree, 0.0)  # members come in the order the sweep cut them off
    assert len(g0) == 1 and g0[0][0] == '-1' and sorted(g0[0][1]) == ['A', 'B', 'C', 'D', 'E']


def _fake_placements(Q, seed, first_bad):
    from apples_amd.engine import F_EXACT, F_PENDANT_INT, F_INSUFFICIENT, F_MISPLACED
    dt = np.dtype([('edge', '<i4'), ('flags', '<u4'), ('error', '<f8'), ('distal', '<f8'), ('pendant', '<f8'),
                   ('n_obs', '<i4'), ('n_valid', '<i4')], align=True)
    rng = np.random.default_rng(seed)
    out = np.zeros(Q, dt)
    out['edge'] = rng.integers(0, 19998, Q)
    out['error'] = rng.random(Q) * 10.0 ** rng.integers(-20, 3, Q)
    out['distal'] = rng.random(Q) * 0.01
    out['pendant'] = rng.random(Q) * 0.01
    kind = rng.integers(0, 10, Q)
    out['flags'][kind == 0] = F_EXACT | F_PENDANT_INT
    out['flags'][kind == 1] = F_PENDANT_INT
    out['flags'][kind == 3] = F_PENDANT_INT | F_MISPLACED
    bad = kind == 2
    out['flags'][bad] = F_INSUFFICIENT | F_PENDANT_INT
    out['edge'][bad] = -1
    if first_bad:
        out['flags'][0] = F_INSUFFICIENT | F_PENDANT_INT
        out['edge'][0] = -1
    return out


@pytest.mark.parametrize('Q', [1, 2, 7, 3000])
@pytest.mark.parametrize('first_bad', [False, True])
def test_streaming_jplace_text_equals_json_dumps(Q, first_bad):
    """The streaming writer and the vectorised row builder produce the bytes of the reference's
    json.dumps(sort_keys=True, indent=4) over join_jplace's dict (run_apples.py:106-118), including
    the keep-first quirk, int/float leakage, exponents and escaped names."""
    from apples_amd.engine import placement_row, placement_rows
    from apples_amd.jplace import dumps, finish, iter_text, join_jplace, keep_mask
    out = _fake_placements(Q, Q + first_bad, first_bad)
    names = ['q%d "x"\\ é' % i if i % 100 == 3 else 'q%d' % i for i in range(Q)]
    tree_string = "('a b':1,b:2.5){1};"
    res = [{'placements': [{'p': [placement_row(p)], 'n': [n]}]} for n, p in zip(names, out)]
    want = dumps(finish(join_jplace(res), tree_string, ['run_apples.py', '-t', 'x y']))
    rows = placement_rows(out)
    assert rows == [placement_row(p) for p in out]
    keep = keep_mask([r[0] for r in rows])
    got = ''.join(iter_text(((n, r) for n, r, k in zip(names, rows, keep) if k), tree_string, ['run_apples.py', '-t', 'x y']))
    assert got == want


def test_worker_rows_rename_exclude_and_messages(capsys):
    """Result assembly without a device: name collision rename, --exclude, the stderr line for
    unplaceable queries (PoolQueryWorker.py:63-70,83-88,119-130)."""
    from apples_amd.options import options_config
    from apples_amd.tree import parse_newick
    from apples_amd.worker import QueryWorker
    tree = parse_newick('((a:1,b:1):1,(c:1,d:1):1);')
    out = _fake_placements(40, 5, True)
    names = ['a' if i == 4 else 'q%d' % i for i in range(40)]
    for exclude in (False, True):
        options, _ = options_config(['-t', 'x', '-s', 'y', '-q', 'z'] + (['--exclude'] if exclude else []))
        w = QueryWorker(tree, options, None, [0])
        got_names, rows = w._rows(names, out)
        assert got_names[4] == 'a-query' and got_names[5] == 'q5'
        from apples_amd.engine import F_MISPLACED, F_EXACT, F_INSUFFICIENT
        mis = ((out['flags'] & F_MISPLACED) != 0) & ((out['flags'] & (F_EXACT | F_INSUFFICIENT)) == 0)
        assert mis.any()
        for i in np.nonzero(mis)[0]:
            assert rows[i][0] == (-1 if exclude else int(out['edge'][i]))
        err = capsys.readouterr().err
        assert err.count('cannot be placed') == int(((out['flags'] & F_INSUFFICIENT) != 0).sum())
        dicts = w._to_jplace(names, out)
        capsys.readouterr()
        assert [d['placements'][0]['p'][0] for d in dicts] == rows


_FASTA_CASES = {
    'plain': b'>a x y\nACGT\n>b\nAC-T\n',
    'multiline': b'junk\n>a\nAC\nGT\n\n>b desc\nacgn\n',
    'no_final_newline': b'>a\nACGT\n>b\nACGTT',
    'crlf': b'>a\r\nACGT\r\n>b\r\nAC-T\r\n',
    'cr_only': b'>a\rACGT\r>b\rAC-T\r',
    'repeated_name': b'>a\nACGT\n>b\nAAAA\n>a\nTTTT\n',
    'fastq': b'@a\nACGT\n+\nIIII\n@b\nAC-T\n+\nII\nII\n',
    'fastq_truncated': b'@a\nACGT\n+\nII\n',
    'fastq_at_in_quality': b'@a\nACGT\n+\n@III\n@b\nAC-T\n+\nIIII\n',
    'headers_only': b'>a\n>b\n',
    'header_last_unterminated': b'>a\nAC\n>',
    'plus_last_unterminated': b'>a\nAC\n+',
    'empty': b'',
    'blank_lines': b'\n\n',
    'ragged': b'>a\nACGT\n>b\nACG\n',
    'empty_name': b'>\nAC\n> x\nGT\n',
    'utf8_name': '>\u00e9\nAC\n>b\nGT\n'.encode(),
    'tab_in_header': b'>a\tb c\nAC\n',
    'other_symbols': b'>a\nA*?.\n>b\nbjou\n',
}


@pytest.mark.parametrize('case', sorted(_FASTA_CASES))
def test_native_fasta_scan_matches_record_reader(case, tmp_path):
    """libapples_io.so's scanner (include/apples_io.h) against the record-by-record restatement of
    apples/fasta2dic.py:4-72, on the reader's quirks: dropped last characters, universal newlines,
    FASTQ blocks, repeated names, ragged rows, both alphabets, masking on and off."""
    from apples_amd import build, fasta
    build.build_io(verbose=False)
    fasta._io_lib = None
    assert fasta._load_io() is not None
    path = str(tmp_path / 'x.fa')
    with open(path, 'wb') as f:
        f.write(_FASTA_CASES[case])
    for prot in (False, True):
        for mask in (False, True):
            try:
                want, werr = fasta._read_alignment_py(path, prot, mask), None
            except Exception as e:  # the reference's failure mode (unequal lengths) must be kept
                want, werr = None, (type(e).__name__, str(e))
            try:
                got, gerr = fasta.read_alignment(path, prot, mask), None
            except Exception as e:
                got, gerr = None, (type(e).__name__, str(e))
            assert werr == gerr
            if want is not None:
                assert want.names == got.names
                assert np.array_equal(want.seqs, got.seqs)


@pytest.mark.parametrize('tail', ['newline', 'none', 'blank'])
def test_threaded_fasta_scan_on_a_file_of_several_megabytes(tail, tmp_path):
    """The threaded scanner (apples_fasta_scan_mt: headers indexed by byte ranges, records filled by record ranges) against the
    general one and the record reader on a 3 MB file: ragged line widths, blank lines inside records, junk before the first
    header, descriptions after the name, lower case and odd symbols, with and without a final newline."""
    import ctypes
    from apples_amd import build, fasta
    build.build_io(verbose=False)
    fasta._io_lib = None
    lib = fasta._load_io()
    rng = np.random.default_rng(5)
    L, n = 1500, 2000
    alpha = np.frombuffer(b'ACGTacgtNnU-*?', dtype=np.uint8)
    seqs = alpha[rng.integers(0, len(alpha), size=(n, L))]
    parts = [b'junk line\n\n']
    for i in range(n):
        parts.append(b'>s%d some description %d\n' % (i, i))
        at = 0
        while at < L:
            w = int(rng.integers(1, 200))
            parts.append(seqs[i, at:at + w].tobytes() + b'\n')
            at += w
            if rng.random() < 0.02:
                parts.append(b'\n')
    blob = b''.join(parts)
    if tail == 'none':
        blob = blob[:-1] + b'G'      # the last line unterminated: it loses its last character, the 'G'
    elif tail == 'blank':
        blob += b'\n\n'
    path = str(tmp_path / 'big.fa')
    with open(path, 'wb') as f:
        f.write(blob)
    assert len(blob) > (1 << 20)
    want = fasta._read_alignment_py(path, False, False)
    got = fasta.read_alignment(path, False, False)
    assert want.names == got.names and np.array_equal(want.seqs, got.seqs)
    # the threaded entry point itself, seven threads, against the general scanner's rows
    data = np.frombuffer(blob, dtype=np.uint8)
    tab = fasta._translation(False, True)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    outs = []
    for fn, extra in ((lib.apples_fasta_scan_mt, (7,)), (lib.apples_fasta_scan, ())):
        n_rec, length, bad, bad_len = ctypes.c_int64(0), ctypes.c_int64(L), ctypes.c_int64(-1), ctypes.c_int64(0)
        mat = np.zeros((n, L), np.uint8)
        off = np.zeros(n, np.int64)
        ln = np.zeros(n, np.int32)
        rc = fn(ptr(data), data.size, ptr(tab), ptr(mat), n, ctypes.byref(n_rec), ctypes.byref(length), ptr(off), ptr(ln),
                ctypes.byref(bad), ctypes.byref(bad_len), *extra)
        assert rc == 0 and n_rec.value == n
        outs.append((mat, off, ln))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    # not the plain shape: the threaded scanner declines
    for odd in (blob.replace(b'\n>s7 ', b'\r\n>s7 ', 1), blob + b'\n+\n', blob + b'\n>'):
        d2 = np.frombuffer(odd, dtype=np.uint8)
        n_rec, length, bad, bad_len = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(-1), ctypes.c_int64(0)
        assert lib.apples_fasta_scan_mt(ptr(d2), d2.size, ptr(tab), None, 0, ctypes.byref(n_rec), ctypes.byref(length), None, None,
                                        ctypes.byref(bad), ctypes.byref(bad_len), 7) == 3


def test_binary_distance_table_equals_text_reader(tmp_path):
    """The .npz form of a -d table gives run_apples.py the same names, columns and values as the
    text form (run_apples.py:43-54), repeated column names included."""
    import run_apples
    rng = np.random.default_rng(3)
    cols = ['a', 'b', 'c', 'b', 'd']
    names = ['q1', 'q2', 'q3']
    D = np.round(rng.random((3, 5)), 6)
    D[1, 2] = -1.0
    text = tmp_path / 'd.mat'
    with open(text, 'w') as f:
        f.write('x ' + ' '.join(cols) + '\n')
        for n, row in zip(names, D):
            f.write(n + ' ' + ' '.join(repr(float(v)) for v in row) + '\n')
    with open(text) as f:
        wn, wc, wD = run_apples.read_dismat(f)
    npz = tmp_path / 'd.npz'
    np.savez(npz, queries=np.array(names), columns=np.array(cols), D=D)
    gn, gc, gD = run_apples.read_dismat_binary(str(npz))
    assert gn == wn and gc == wc
    assert np.array_equal(gD, wD)


def test_database_cache_round_trip(tmp_path):
    """The .npz database holds exactly what a run would have built from the tree and the alignment:
    tree arrays, extended Newick, alignment rows, clusters and consensus rows."""
    import build_applesdtb
    from apples_amd import database, treecluster
    from apples_amd.fasta import read_alignment
    from apples_amd.reference import ReducedReference
    from apples_amd.tree import extended_newick, read_tree
    db = str(tmp_path / 'c1.dtb')
    build_applesdtb.main(['-s', os.path.join(DATA, 'ref.fa'), '-t', os.path.join(DATA, 'backbone.nwk'), '-o', db, '-f', '0.2', '-D'])
    tree, newick, ref, protein, thr = database.load(db)
    t0 = read_tree(os.path.join(DATA, 'backbone.nwk'))
    a0 = read_alignment(os.path.join(DATA, 'ref.fa'), False, False)
    r0 = ReducedReference(a0, False, treecluster.grouped(t0, 0.2 * 1.2))
    assert protein is False and thr == 0.2 and newick == extended_newick(t0)
    for f in ('parent', 'edge_len', 'has_len', 'child_off', 'child_idx', 'level'):
        assert np.array_equal(getattr(tree, f), getattr(t0, f)), f
    assert tree.labels == t0.labels and tree.is_rooted == t0.is_rooted and tree.name_to_node == t0.name_to_node
    assert ref.aln.names == a0.names and np.array_equal(ref.aln.seqs, a0.seqs)
    for got, want in zip(ref.cluster_arrays(), r0.cluster_arrays()):
        assert np.array_equal(got, want)
    with pytest.raises(ValueError):
        database.load(_not_a_database(tmp_path))


def _not_a_database(tmp_path):
    p = str(tmp_path / 'other.npz')
    np.savez(p, x=np.arange(3))
    return p


def _parse_both(text, monkeypatch):
    """parse_newick through the native scanner and through the Python token loop."""
    from apples_amd import tree as T
    out = []
    for native in (True, False):
        with monkeypatch.context() as m:
            if not native:
                m.setattr(T, '_scan_native', lambda t: None)
            try:
                out.append(('ok', T.parse_newick(text)))
            except Exception as e:  # noqa: BLE001 -- the two paths must fail alike, whatever the failure
                out.append(('err', type(e).__name__, str(e)))
    return out


def _same_tree(a, b):
    if a[0] != b[0]:
        return False
    if a[0] == 'err':
        return a[1:] == b[1:]
    x, y = a[1], b[1]
    return (np.array_equal(x.parent, y.parent) and np.array_equal(x.edge_len, y.edge_len, equal_nan=True)
            and np.array_equal(x.has_len, y.has_len) and x.name_to_node == y.name_to_node and x.labels == y.labels
            and np.array_equal(x.child_off, y.child_off) and np.array_equal(x.child_idx, y.child_idx)
            and np.array_equal(x.level, y.level) and x.is_rooted == y.is_rooted)


NEWICK_CASES = [
    "(A:1,B:2);", "[&R] ((A:0.1,B:0.2)X:0.3,C:1e-3)root;", "(A,B,(C,D)E)F;", "('it''s':1,'b c':2)'r';",
    "(A:1,B:2", "(A:1,,B:2);", "(A:,B:1);", "(A:1[c],B[comment]:2)[x];", "(A:1,B:2));", "A;", ";", "",
    "(A:nan,B:inf);", "(A:1_0,B:0x10);", "(A:abc,B:def);", "( A : 1.5 , B :\t2 ) ;", "(A:1,B:2)C:3;D",
    "('a:1,b:2);", "(a]b:1,c[d:2);", "(A:'1',B:2);", "(A:+.5,B:-1.E3,C:1.,D:.e1);", "('''':1,'''a':2);",
    "(A:1,'b'c:2);", "(A B:1, C  D :2);", "(A:1\n,\nB:2\n)\n;\n", "((((A))));", "(,);", "(:1,:2):3;",
    "(A:1e400,B:1e-400,C:00012.5000);", "(A:1,B:2);(C:1,D:2);", "[&U](A:1,B:2);", "(A:1:2,B:3);", "(A::1,B:3);",
    "(a'b':1,c:2);", "(\x1fA\x1f:1,\x0bB:2);", "(A:1,B:2)é;", "(A:1e-:8,B:2);", "(A:1x:2,B:zz);", "(''':1,B:2);",
]


@pytest.mark.parametrize('text', NEWICK_CASES)
def test_native_newick_scanner_equals_python_token_loop_on_odd_inputs(text, monkeypatch):
    """include/apples_io.h:apples_newick_scan against the token loop of apples_amd/tree.py (SURVEY
    Appendix B contract): quoted labels with doubled quotes, comments, stray brackets and quotes,
    white space, missing / repeated / non-decimal branch lengths, polytomies, unbalanced input,
    text after the semicolon, non-ASCII labels -- same arrays, or the same error with the same words."""
    a, b = _parse_both(text, monkeypatch)
    assert _same_tree(a, b), (text, a[:3], b[:3])


def test_native_newick_scanner_equals_python_token_loop_on_random_trees(monkeypatch):
    import random
    rng = random.Random(5)

    def label():
        k = rng.random()
        if k < 0.3:
            return ''
        if k < 0.5:
            return "'q %d''x'" % rng.randrange(100)
        if k < 0.6:
            return ' t%d ' % rng.randrange(1000)
        return 't%d' % rng.randrange(100000)

    def length():
        k = rng.random()
        if k < 0.15:
            return ''
        if k < 0.25:
            return ':%d' % rng.randrange(10)
        if k < 0.3:
            return ': %.3e ' % rng.random()
        if k < 0.32:
            return ':-0.5'
        return ':' + repr(rng.random() * 10 ** rng.randrange(-8, 3))

    def subtree(depth=0):
        comment = '[c%d]' % rng.randrange(9) if rng.random() < 0.05 else ''
        if depth > 12 or rng.random() < 0.35:
            return label() + comment + length()
        k = 2 if rng.random() < 0.8 else rng.randrange(1, 6)
        return '(' + ','.join(subtree(depth + 1) for _ in range(k)) + ')' + label() + length()

    for _ in range(800):
        text = subtree() + ';'
        if rng.random() < 0.15:  # one character replaced: mostly malformed, sometimes just different
            j = rng.randrange(len(text))
            text = text[:j] + rng.choice("(),:;'[]x ") + text[j + 1:]
        a, b = _parse_both(text, monkeypatch)
        assert _same_tree(a, b), (text[:300], a[:3], b[:3])


def test_native_clustering_sweep_equals_the_python_sweep():
    """include/apples_io.h:apples_max_clusters against apples_amd/treecluster.py:_max_clusters_py on random
    trees with polytomies, unifurcations, zero and missing lengths, at thresholds from "every leaf alone"
    to "one cluster": the same clusters, in the same order, leaves in the same order."""
    import random
    from apples_amd import treecluster as C
    from apples_amd.tree import parse_newick
    rng = random.Random(9)

    def length():
        k = rng.random()
        if k < 0.1:
            return ''
        if k < 0.15:
            return ':0'
        return ':' + repr(rng.random() * rng.choice([0.001, 0.01, 0.1, 1.0]))

    def subtree(depth=0):
        if depth > 10 or rng.random() < 0.3:
            return 't%d' % rng.randrange(10 ** 9) + length()
        r = rng.random()
        k = 2 if r < 0.7 else (1 if r < 0.75 else rng.randrange(3, 7))
        return '(' + ','.join(subtree(depth + 1) for _ in range(k)) + ')' + length()

    checked = 0
    for _ in range(300):
        tree = parse_newick(subtree() + ';')
        for thr in (0.0, 0.05, 0.24, 1.0, 100.0):
            native = C._max_clusters_native(tree, thr)
            if native is None:
                pytest.skip('libapples_io.so not built')
            assert native == C._max_clusters_py(tree, thr)
            checked += 1
    assert checked == 1500


def test_database_flag_combinations_follow_the_reference(caplog):
    """apples/OptionsRun.py:88-107: -a with -s is an error; -a with -t or -d is accepted with a warning
    (the user's tree wins, database sequences are ignored); -a alone needs no -t."""
    import logging
    with pytest.raises(ValueError):
        options_config(['-a', 'db', '-s', 'r.fa', '-q', 'q.fa'])
    with caplog.at_level(logging.WARNING):
        o, _ = options_config(['-a', 'db', '-t', 'x.nwk', '-q', 'q.fa'])
        assert o.tree_fp == 'x.nwk' and o.database_fp == 'db'
        assert 'User provided tree has higher priority' in caplog.text
        caplog.clear()
        o, _ = options_config(['-a', 'db', '-d', 'd.mat'])
        assert o.reestimate_backbone is False
        assert 'Database sequences will be ignored' in caplog.text
    o, _ = options_config(['-a', 'db', '-q', 'q.fa'])
    assert o.tree_fp is None
    with pytest.raises(ValueError):
        options_config(['-q', 'q.fa'])


def test_reference_alignment_may_hold_rows_beyond_the_clusters(tmp_path):
    """The reference builds representatives from TreeCluster's output (backbone leaves) alone
    (apples/Reference.py:94-107): rows of -s that are not backbone leaves are never compared, but
    their names still keep them out of the query set of an extended alignment (run_apples.py:85-89).
    ReducedReference keeps the full alignment for names and a clustered subset for the device."""
    from apples_amd import database, treecluster
    from apples_amd.fasta import Alignment
    from apples_amd.tree import extended_newick, read_tree
    tree = read_tree(os.path.join(DATA, 'backbone.nwk'))
    a0 = read_alignment(os.path.join(DATA, 'ref.fa'), False, False)
    extra = np.full((2, a0.length), ord('A'), np.uint8)
    # two extra rows, one in the middle, one at the end
    names = a0.names[:100] + ['not_in_tree_1'] + a0.names[100:] + ['not_in_tree_2']
    seqs = np.vstack([a0.seqs[:100], extra[:1], a0.seqs[100:], extra[1:]])
    sup = Alignment(names, seqs)
    clusters = treecluster.grouped(tree, 0.2 * 1.2)
    r0 = ReducedReference(a0, False, clusters)
    r1 = ReducedReference(sup, False, clusters)
    assert r0.eng_rows is None and r0.eng_aln is r0.aln
    assert len(r1.aln) == len(a0) + 2 and 'not_in_tree_1' in r1.aln.index
    assert r1.eng_aln.names == a0.names and np.array_equal(r1.eng_aln.seqs, a0.seqs)
    for got, want in zip(r1.cluster_arrays(), r0.cluster_arrays()):
        assert np.array_equal(got, want)
    # the database cache keeps both views
    db = str(tmp_path / 'sup.dtb')
    database.save(db, tree, extended_newick(tree), r1, 0.2)
    _, _, r2, _, _ = database.load(db)
    assert r2.aln.names == names and r2.eng_aln.names == a0.names
    for got, want in zip(r2.cluster_arrays(), r0.cluster_arrays()):
        assert np.array_equal(got, want)
    # a sequence listed twice is still an error; a cluster member without a sequence is a KeyError
    with pytest.raises(ValueError):
        ReducedReference(a0, False, [('-1', [a0.names[0], a0.names[0]])])
    with pytest.raises(KeyError):
        ReducedReference(a0, False, [('-1', ['no_such_sequence'])])


@pytest.mark.parametrize('text', [
    '\tA B C\nq1 0.1 0.2 0.3\nq2 1e-3 -1 2.5E+1\n',
    'x A B A\r\nq 1 2 3\r\n  r\t4\t5\t6  \r\n',                 # first header field dropped, duplicate tag, CRLF, padding
    ' A B C D\nq 1 2\n\nlast 7 8 9 10 11 12',                   # short row, empty line, surplus values, no final newline
    '\tA B\nq nan 1\n',                                          # not a plain decimal: the Python reader decides
    '\tA B\nq 1_0 2\n',
    '\tA\n',
    '',
])
def test_native_distance_table_scanner_equals_the_python_reader(text, tmp_path):
    from apples_amd import dismat
    p = tmp_path / 't.mat'
    with open(p, 'w', newline='') as f:
        f.write(text)
    with open(p) as f:
        try:
            want = dismat.read_dismat_py(f)
        except Exception as e:  # whatever the reference's reader raises, the front end raises too
            with pytest.raises(type(e)):
                dismat.read_dismat_text(str(p))
            return
    got = dismat.read_dismat_text(str(p))
    assert got[0] == want[0] and got[1] == want[1]
    assert got[2].shape == want[2].shape and np.array_equal(got[2], want[2], equal_nan=True)


def test_native_distance_table_scanner_on_the_reference_tables_and_at_size(tmp_path):
    from apples_amd import dismat
    from apples_amd.fasta import _load_io
    assert hasattr(_load_io(), 'apples_dismat_scan')
    for name in ('dist.mat', 'small_dist.mat'):
        with open(os.path.join(DATA, name)) as f:
            want = dismat.read_dismat_py(f)
        got = dismat.read_dismat(os.path.join(DATA, name))
        assert got[0] == want[0] and got[1] == want[1] and np.array_equal(got[2], want[2])
    rng = np.random.default_rng(3)
    D = np.round(rng.uniform(0, 2, size=(40, 3000)), 8)
    D[3, ::7] = -1.0
    p = tmp_path / 'big.mat'
    with open(p, 'w') as f:
        f.write('\t' + ' '.join('c%d' % i for i in range(3000)) + '\n')
        for i in range(40):
            f.write('q%d ' % i + ' '.join(repr(float(v)) for v in D[i]) + '\n')
    names, cols, got = dismat.read_dismat(str(p))
    assert names == ['q%d' % i for i in range(40)] and cols == ['c%d' % i for i in range(3000)]
    assert np.array_equal(got, D)


def test_native_distance_table_values_equal_python_float_on_random_spellings(tmp_path):
    """The scanner's exact fast path (at most 15 significant digits, power of ten within 10^22) and its
    strtod path against Python's float() on 30 000 random decimal spellings."""
    import random
    from apples_amd import dismat
    random.seed(7)
    vals = []
    for _ in range(30000):
        k = random.randint(1, 18)
        digs = ''.join(random.choice('0123456789') for _ in range(k))
        pos = random.randint(0, k)
        s = digs[:pos] + '.' + digs[pos:] if random.random() < 0.8 else digs
        if s == '.':
            s = '0.'
        if random.random() < 0.3:
            s += 'e%+d' % random.randint(-30, 30)
        if random.random() < 0.2:
            s = '-' + s
        vals.append(s)
    p = tmp_path / 's.mat'
    with open(p, 'w') as f:
        f.write('\t' + ' '.join('c%d' % i for i in range(len(vals))) + '\n')
        f.write('q ' + ' '.join(vals) + '\n')
    _, _, G = dismat.read_dismat(str(p))
    assert np.array_equal(G[0], np.array([float(v) for v in vals]))


def test_native_jplace_rows_equal_the_python_writer():
    """apples_jplace_rows (include/apples_io.h) against apples_amd.jplace.iter_text, itself byte-identical to the reference's
    json.dumps(sort_keys=True, indent=4) (run_apples.py:116): floats spelled as Python's repr, the three row kinds, dropped rows,
    the first-result quirk of join_jplace (apples/jutil.py:11-18), names that need escaping (the native writer declines)."""
    import io
    import json
    from apples_amd import build, fasta, jplace
    build.build_io(verbose=False)
    fasta._io_lib = None
    rng = np.random.default_rng(1)
    n = 3000
    names = ['q%d' % i for i in range(n)]
    edge = rng.integers(-1, 1000, size=n).astype(np.int32)
    kind = rng.integers(0, 3, size=n).astype(np.uint8)
    err = rng.random(n) * 10.0 ** rng.integers(-25, 20, size=n)
    err[:8] = [0.0, -0.0, 1e16, 1e-4, 9.999e-5, 123456789012345680.0, 5e-324, 1.7976931348623157e308]
    dist = rng.random(n)
    pend = rng.random(n) * 1e-5
    for first_edge in (-1, 7):
        edge[0] = first_edge
        cols = {'edge': edge, 'error': err, 'distal': dist, 'pendant': pend, 'kind': kind}
        rows = [[int(e), 0, 1, 0, 0] if k == 1 else [int(e), float(a), 1, float(b), 0 if k == 2 else float(c)]
                for e, a, b, c, k in zip(edge, err, dist, pend, kind)]
        keep = jplace.keep_mask([r[0] for r in rows])
        want = ''.join(jplace.iter_text(((nm, r) for nm, r, k in zip(names, rows, keep) if k), '(a{0},b{1});', ['run', 'a b']))
        f = io.BytesIO()
        assert jplace.write_native(f, names, cols, '(a{0},b{1});', ['run', 'a b'])
        assert f.getvalue().decode() == want
        assert json.loads(want)['version'] == 3
    one = {k: v[:1] for k, v in cols.items()}
    for bad in ('a"b', 'a\\b', 'caf\u00e9', 'tab\there'):
        assert not jplace.write_native(io.BytesIO(), [bad], one, 't', ['x'])
    f = io.BytesIO()
    assert jplace.write_native(f, [], {k: v[:0] for k, v in cols.items()}, 't', ['x'])
    assert f.getvalue().decode() == ''.join(jplace.iter_text([], 't', ['x']))


def test_native_float_spelling_is_pythons():
    import ctypes
    import json
    import random
    import struct
    from apples_amd import build, fasta
    build.build_io(verbose=False)
    fasta._io_lib = None
    lib = fasta._load_io()
    lib.apples_format_double.restype = ctypes.c_int64
    lib.apples_format_double.argtypes = [ctypes.c_double, ctypes.c_char_p]
    buf = ctypes.create_string_buffer(64)
    rng = random.Random(7)
    vals = [0.0, -0.0, 1.0, 1e16, 9999999999999998.0, 1e-4, 9.999e-5, 1e22, 5e-324, 0.1, 1 / 3, 100.0, 1e-5, float('nan'), float('inf'), float('-inf')]
    vals += [struct.unpack('d', struct.pack('Q', rng.getrandbits(64)))[0] for _ in range(50000)]
    vals += [rng.random() * 10 ** rng.randint(-8, 20) for _ in range(20000)]
    for v in vals:
        k = lib.apples_format_double(v, buf)
        assert buf.raw[:k].decode() == json.dumps(v), repr(v)


def test_native_extended_newick_and_consensus_equal_python():
    """apples_extended_newick against the Python formatter (apples/jutil.py:22-96: integral lengths as ints, others as str(float),
    no length where the input had none, quoted labels as parsed) and apples_consensus against the numpy restatement of
    apples/PoolRepresentativeWorker.py:17-85 (ties to the first symbol in alphabet order, other symbols not counted)."""
    from apples_amd import build, fasta, reference, synth
    from apples_amd import tree as T
    build.build_io(verbose=False)
    fasta._io_lib = None
    for nw in ('((A:0.1,B:2):3,(C:1e-7,(D:0.25,E)x:-0.5)y:17,F:5e-324)r;', '[&R] (a,b,(c,d)e);', 'A;', "('a b':1,'c''d':2.50):0;",
               '((a:1.0,b:nan):inf,c:-0.0);', synth.random_tree_newick(3000)):
        t = T.parse_newick(nw)
        assert T._extended_newick_native(t) == T._extended_newick_py(t)
    big = T.parse_newick('((A:0.1,B:2):3,C:1e22);')   # an integral length beyond 2^63: left to Python
    assert T._extended_newick_native(big) is None and T.extended_newick(big) == T._extended_newick_py(big)
    rng = np.random.default_rng(2)
    for protein in (False, True):
        alpha = np.frombuffer(b'ACGT-NX*acgt' if not protein else b'ACDEFGHIKLMNPQRSTVWY-XBZ*', np.uint8)
        seqs = alpha[rng.integers(0, len(alpha), size=(2000, 131))]
        mrow = list(rng.permutation(2000))
        groups, at = [], 0
        while at < 2000:
            k = int(rng.integers(1, 60))
            groups.append((at, min(2000, at + k)))
            at += k
        got = reference._consensus_rows(seqs, mrow, groups, protein)
        want = np.array([reference.consensus(seqs[mrow[a:b]], protein) for a, b in groups], np.uint8)
        assert np.array_equal(got, want)


def test_synthetic_backbone_shapes_keep_the_leaves_and_their_root_distances():
    """apples_amd/synth.py: the shapes bench.py's `other_shapes` legs and the GPU fuzz draw beside the strictly binary random-join
    tree -- unrooted (the root's first internal child dissolved: a trifurcation), polytomies (a share of the internal nodes dissolved
    into their parents) and a caterpillar spine (more than `spine` levels).  Same leaves in the same left-to-right order, same root
    distance of every leaf (the branch lengths print with six decimals, sums of two such numbers print exactly)."""
    from apples_amd import synth
    t = synth.parse_newick(synth.random_tree_newick(400, seed=3))
    assert np.diff(t.child_off).max() == 2

    def root_dist(tr):
        r = np.zeros(tr.n_nodes)
        for v in range(tr.n_nodes - 2, -1, -1):
            r[v] = r[tr.parent[v]] + tr.edge_len[v]
        return {tr.labels[v]: r[v] for v in tr.leaves}

    base = root_dist(t)
    u = synth.reshape_tree(t, 'unrooted')
    assert u.n_nodes == t.n_nodes - 1 and len(u.children(u.root)) == 3 and np.diff(u.child_off).max() == 3
    p = synth.reshape_tree(t, 'polytomies', seed=5, frac=0.3)
    assert p.n_nodes < t.n_nodes - 50 and np.diff(p.child_off).max() >= 4
    for x in (u, p):
        assert [x.labels[v] for v in x.leaves] == [t.labels[v] for v in t.leaves]
        rd = root_dist(x)
        assert max(abs(rd[k] - base[k]) for k in base) < 1e-12
        assert (np.diff(x.child_off)[~x.is_leaf] >= 2).all()
    d = synth.make_dataset(900, 40, 5, spine=300)
    assert d.tree.level.max() > 300 and np.diff(d.tree.child_off).max() == 2 and d.tree.n_leaves == 900
    # the default generator's stream is what it was (fixtures and benchmark inputs depend on it)
    import hashlib
    assert hashlib.sha1(synth.random_tree_newick(50, 1).encode()).hexdigest()[:12] == '51bad92e7a9d'
