"""ctypes binding of libapples_hip.so -- the only way the Python host reaches the GPU.

There is no CPU fallback: if the library is missing or no MI355X is visible the
constructor raises.  Declarations mirror include/apples_hip.h.
"""
import ctypes as C
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libapples_hip.so')

METHODS = {'OLS': 0, 'FM': 1, 'BME': 2, 'BE': 3}
CRITERIA = {'MLSE': 0, 'ME': 1, 'HYBRID': 2}

F_EXACT, F_INSUFFICIENT, F_MISPLACED, F_PENDANT_INT, F_ZERO_NOT_IN_TREE, F_DEGENERATE = 1, 2, 4, 8, 16, 32
# apples_params.debug (include/apples_hip.h APPLES_DBG_*): alternative routes to the same placements, fixed at context creation
DBG = {'no_fuse': 1, 'sweep_scan': 2, 'node_map': 4, 'sweep_merge': 8, 'no_sweep_merge': 16, 'no_dist_gemm': 32, 'no_sweep_lean': 64,
       'no_sd_gemm': 128, 'cluster_by_query': 256, 'no_cluster_topup': 512, 'no_stream_select': 1024, 'no_topup_kernel': 2048,
       'no_cluster_big': 4096, 'no_sd_topup': 8192, 'sd_fp6': 16384, 'no_topup_overlap': 32768, 'stream_third_pass': 65536, 'no_sd_compact': 131072, 'sd_compact_tiny': 262144, 'no_blocks': 524288, 'hybrid_records': 1048576, 'no_cluster_mfma': 2097152}
T_PACK, T_DIST, T_SELECT, T_SWEEP, T_TOTAL, T_DIST_LAUNCHES, T_FILTER, T_BLOCKS, T_COUNT = range(9)

PLACEMENT_DTYPE = np.dtype([('edge', '<i4'), ('flags', '<u4'), ('error', '<f8'), ('distal', '<f8'),
                            ('pendant', '<f8'), ('n_obs', '<i4'), ('n_valid', '<i4')], align=True)

EXPORTS = ['apples_ctx_create', 'apples_ctx_destroy', 'apples_last_error', 'apples_set_params', 'apples_distances',
           'apples_place_from_sequences', 'apples_place_from_distances', 'apples_sweep_edges', 'apples_place_sequences_streamed',
           'apples_queries_upload', 'apples_table_upload', 'apples_queries_free', 'apples_place_resident', 'apples_fetch_placements',
           'apples_distances_resident', 'apples_placements_device_ptr', 'apples_last_timing', 'apples_describe', 'apples_device_log',
           'apples_backbone_lengths', 'apples_abi_version', 'apples_params_size']
ABI_VERSION = 8  # include/apples_hip.h APPLES_ABI_VERSION


class _Tree(C.Structure):
    _fields_ = [('n_nodes', C.c_int32), ('parent', C.c_void_p), ('edge_len', C.c_void_p), ('child_off', C.c_void_p),
                ('child_idx', C.c_void_p), ('level', C.c_void_p)]


class _Alignment(C.Structure):
    _fields_ = [('n_rows', C.c_int64), ('n_refs', C.c_int64), ('length', C.c_int32), ('rows', C.c_void_p),
                ('row_node', C.c_void_p), ('n_reps', C.c_int64), ('rep_row', C.c_void_p), ('member_off', C.c_void_p),
                ('member_row', C.c_void_p)]


class _Params(C.Structure):
    _fields_ = [('model', C.c_int32), ('method', C.c_int32), ('criterion', C.c_int32), ('negative_branch', C.c_int32),
                ('filt_threshold', C.c_double), ('base_observation', C.c_int32), ('overlap_frac', C.c_double),
                ('jc_lut', C.c_void_p), ('jc_lut_len', C.c_int64), ('max_batch', C.c_int64), ('debug', C.c_uint32),
                ('batch_gib', C.c_int32), ('knobs', C.c_char_p)]


_lib = None


def load_library():
    """dlopen the in-tree library; raises with a build hint when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libapples_hip.so is not built (%s); run `python -m apples_amd.build`. '
                           'The APPLES hot path has no CPU fallback.' % LIB_PATH)
    # the sweep runs its launches side by side on streams of their own; the HIP runtime gives a process four hardware queues
    # by default and lets further streams share them, serialised (csrc/api.hip:apples_ctx_create, scripts/stream_queue_probe.hip).
    # Room for other users of the process (read when the runtime starts: no effect if something has initialised HIP already,
    # none if the user set it)
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    lib = C.CDLL(LIB_PATH)
    lib.apples_last_error.restype = C.c_char_p
    lib.apples_last_error.argtypes = [C.c_void_p]
    lib.apples_describe.restype = C.c_char_p
    lib.apples_describe.argtypes = [C.c_void_p]
    lib.apples_ctx_create.argtypes = [C.POINTER(_Tree), C.c_void_p, C.POINTER(_Params), C.c_int, C.POINTER(C.c_void_p)]
    lib.apples_ctx_destroy.argtypes = [C.c_void_p]
    lib.apples_ctx_destroy.restype = None
    lib.apples_set_params.argtypes = [C.c_void_p, C.POINTER(_Params)]
    lib.apples_distances.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.apples_place_from_sequences.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.apples_place_from_distances.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                                C.c_void_p]
    lib.apples_sweep_edges.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 7
    lib.apples_queries_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]
    lib.apples_place_sequences_streamed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]
    lib.apples_table_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_int64)]
    lib.apples_queries_free.argtypes = [C.c_void_p, C.c_int64]
    lib.apples_place_resident.argtypes = [C.c_void_p, C.c_int64]
    lib.apples_fetch_placements.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    lib.apples_distances_resident.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    lib.apples_placements_device_ptr.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    lib.apples_last_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.apples_backbone_lengths.argtypes = [C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int64, C.c_int32, C.c_int, C.c_int64, C.c_void_p]
    lib.apples_device_log.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    # the mirror structs here must be the library's: a library built from another header is refused, not guessed at
    lib.apples_abi_version.restype = C.c_uint32
    lib.apples_params_size.restype = C.c_size_t
    if lib.apples_abi_version() != ABI_VERSION or lib.apples_params_size() != C.sizeof(_Params):
        raise RuntimeError('libapples_hip.so has ABI %d / apples_params of %d bytes; this engine.py expects %d / %d: rebuild '
                           '(`python -m apples_amd.build --force`)' % (lib.apples_abi_version(), lib.apples_params_size(),
                                                                        ABI_VERSION, C.sizeof(_Params)))
    _lib = lib
    return lib


def jc69_lut(length, overlap_frac):
    """Distance for every integer pair (mism, valid), valid in [0, L], laid out
    [valid*(valid+1)/2 + mism]; evaluated with the reference's own expression order and numpy's
    log (apples/distance.py:734-745) so the device hands out the reference's bits."""
    L = int(length)
    valid = np.repeat(np.arange(L + 1, dtype=np.int64), np.arange(1, L + 2))
    start = valid * (valid + 1) // 2
    mism = np.arange(len(valid), dtype=np.int64) - start
    out = np.full(len(valid), -1.0)
    with np.errstate(all='ignore'):
        ok = (valid > 0) & ~((valid / L) < overlap_frac)
        p = np.zeros(len(valid))
        p[ok] = mism[ok] * 1.0 / valid[ok]
        zero = ok & (p - np.finfo(float).eps < 0)
        loc = 1 - (4 * p / 3)
        pos = ok & ~zero & ~(0 >= loc)
        out[zero] = 0.0
        out[pos] = -0.75 * np.log(loc[pos])
    return out


def backbone_lengths(parent, children, leaf_row, rows, protein, device=0, site_chunk=0):
    """Minimum-evolution branch lengths of a fixed topology on the GPU (apples_backbone_lengths; what the
    reference gets from FastTree, apples/reestimateBackbone.py:82-84).  parent[v] (-1 root), children[v] lists,
    leaf_row[v] row of `rows` (uint8 [n_rows, L], FASTA bytes) for leaves.  Returns float64[n_nodes]."""
    lib = load_library()
    n = len(parent)
    par = np.ascontiguousarray(parent, dtype=np.int32)
    off = np.zeros(n + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(c) for c in children])
    idx = np.ascontiguousarray([c for cs in children for c in cs], dtype=np.int32)
    lrow = np.ascontiguousarray(leaf_row, dtype=np.int32)
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    if rows.ndim != 2:
        raise ValueError('rows must be a [n_rows, L] byte matrix')
    out = np.zeros(n, dtype=np.float64)
    rc = lib.apples_backbone_lengths(int(device), n, _ptr(par), _ptr(off), _ptr(idx), _ptr(lrow), _ptr(rows), rows.shape[0],
                                     rows.shape[1], int(bool(protein)), int(site_chunk), _ptr(out))
    if rc != 0:
        raise RuntimeError(lib.apples_last_error(None).decode())
    return out


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def device_log(x, device=0):
    """The distance kernels' logarithm on an array of positive doubles (apples_device_log: libm's log bit for bit)."""
    lib = load_library()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    if lib.apples_device_log(int(device), _ptr(x), x.size, _ptr(out)) != 0:
        raise RuntimeError(lib.apples_last_error(None).decode())
    return out


class Engine:
    """One device context: resident tree + packed reference alignment + workspaces."""

    LUT_MAX_LEN = 8192  # 33.6 M entries (268 MB) at most; longer alignments use the device log

    def __init__(self, tree, ref_seqs=None, ref_nodes=None, clusters=None, protein=False, method='FM',
                 criterion='MLSE', negative=False, threshold=0.2, baseobs=25, overlap=0.001, device=0,
                 use_lut=True, max_batch=0, debug=(), batch_gib=0, knobs=None):
        """tree: apples_amd.tree.Tree.  ref_seqs: uint8[N, L] (None for a distance-table context).
        ref_nodes: int32[N] tree leaf of each row (-1 = not in tree).  clusters: None (all
        singletons) or (cons_rows uint8[C, L], rep_row int32[R], member_off int32[R+1], member_row).
        batch_gib: cap of the device batch buffers (0 = sized from free memory alone; the cap only lowers that).
        knobs: {name: value} tuning / test knobs of this context alone (apples_params.knobs; the process environment's
        APPLES_* variables give the same knobs to every context)."""
        self.lib = load_library()
        self.tree = tree
        self._keep = []
        t = _Tree()
        t.n_nodes = tree.n_nodes
        arrs = [np.ascontiguousarray(tree.parent, np.int32), np.ascontiguousarray(tree.edge_len, np.float64),
                np.ascontiguousarray(tree.child_off, np.int32), np.ascontiguousarray(tree.child_idx, np.int32),
                np.ascontiguousarray(tree.level, np.int32)]
        self._keep += arrs
        t.parent, t.edge_len, t.child_off, t.child_idx, t.level = [_ptr(a) for a in arrs]
        self.protein = bool(protein)
        self.length = 0
        self.n_refs = 0
        self.n_rows = 0
        aln_p = None
        if ref_seqs is not None:
            ref_seqs = np.ascontiguousarray(ref_seqs, np.uint8)
            self.n_refs, self.length = ref_seqs.shape
            rows = ref_seqs
            a = _Alignment()
            if clusters is not None:
                cons, rep_row, member_off, member_row = clusters
                cons = np.ascontiguousarray(cons, np.uint8).reshape(-1, self.length)
                rows = np.ascontiguousarray(np.vstack([ref_seqs, cons])) if len(cons) else ref_seqs
                rep_row = np.ascontiguousarray(rep_row, np.int32)
                member_off = np.ascontiguousarray(member_off, np.int32)
                member_row = np.ascontiguousarray(member_row, np.int32)
                self._keep += [rep_row, member_off, member_row]
                a.n_reps = len(rep_row)
                a.rep_row, a.member_off, a.member_row = _ptr(rep_row), _ptr(member_off), _ptr(member_row)
            ref_nodes = np.ascontiguousarray(ref_nodes, np.int32)
            self._keep += [rows, ref_nodes]
            self.n_rows = rows.shape[0]
            a.n_rows, a.n_refs, a.length = rows.shape[0], self.n_refs, self.length
            a.rows, a.row_node = _ptr(rows), _ptr(ref_nodes)
            aln_p = C.cast(C.pointer(a), C.c_void_p)
            self._keep.append(a)
        self.use_lut = bool(use_lut)
        self.max_batch = int(max_batch)
        self.batch_gib = int(batch_gib)
        self.knobs = ';'.join('%s=%d' % (k, int(v)) for k, v in (knobs or {}).items()).encode() or None
        self.debug = sum(DBG[k] for k in debug) if not isinstance(debug, int) else int(debug)
        self._opts = dict(method=method, criterion=criterion, negative=negative, threshold=threshold, baseobs=baseobs,
                          overlap=overlap)
        p = self._params()
        ctx = C.c_void_p()
        rc = self.lib.apples_ctx_create(C.byref(t), aln_p, C.byref(p), int(device), C.byref(ctx))
        if rc != 0:
            raise RuntimeError('apples_ctx_create failed: %s' % self.lib.apples_last_error(None).decode())
        self.ctx = ctx

    # ------------------------------------------------------------------ options
    def _params(self):
        o = self._opts
        p = _Params()
        p.model = 1 if self.protein else 0
        p.method = METHODS.get(o['method'], 0)          # anything else means OLS (PoolQueryWorker.py:104-111)
        p.criterion = CRITERIA.get(o['criterion'], 0)   # anything else means MLSE (Algorithm.py:76-91)
        p.negative_branch = 1 if o['negative'] else 0
        p.filt_threshold = float(o['threshold'])
        p.base_observation = int(o['baseobs'])
        p.overlap_frac = float(o['overlap'])
        p.max_batch = self.max_batch
        p.debug = self.debug
        p.batch_gib = self.batch_gib
        p.knobs = self.knobs
        self._lut = None
        if not self.protein and self.use_lut and 0 < self.length <= self.LUT_MAX_LEN:
            self._lut = jc69_lut(self.length, float(o['overlap']))
            p.jc_lut = _ptr(self._lut)
            p.jc_lut_len = len(self._lut)
        return p

    def set_options(self, **kw):
        """Change method / criterion / negative / threshold / baseobs / overlap on the resident context."""
        self._opts.update(kw)
        p = self._params()
        self._check(self.lib.apples_set_params(self.ctx, C.byref(p)))

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError('libapples_hip: %s' % self.lib.apples_last_error(self.ctx).decode())

    def close(self):
        if getattr(self, 'ctx', None):
            self.lib.apples_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def describe(self):
        return json.loads(self.lib.apples_describe(self.ctx).decode())

    def timing(self):
        ms = np.zeros(T_COUNT)
        self.lib.apples_last_timing(self.ctx, _ptr(ms), T_COUNT)
        return {'pack_ms': ms[T_PACK], 'dist_ms': ms[T_DIST], 'select_ms': ms[T_SELECT], 'sweep_ms': ms[T_SWEEP],
                'total_ms': ms[T_TOTAL], 'dist_launches': int(ms[T_DIST_LAUNCHES]), 'filter_ms': ms[T_FILTER], 'blocks_ms': ms[T_BLOCKS]}

    def _queries(self, queries):
        """uint8[Q, L] with L = the reference alignment's length.  The reference fails loudly on a query
        of another length (numpy elementwise compare, apples/distance.py:733); so does this: bytes are
        never re-chunked into a different number of queries."""
        if self.length <= 0:
            raise ValueError('this context has no reference alignment (distance-table context)')
        q = np.asarray(queries)
        if q.ndim == 1 and q.size == self.length:
            q = q.reshape(1, -1)
        if q.ndim != 2 or q.shape[1] != self.length:
            raise ValueError('query alignment must be a [Q, %d] byte matrix (reference alignment length %d), got shape %s'
                             % (self.length, self.length, q.shape))
        return np.ascontiguousarray(q, np.uint8)

    # ------------------------------------------------------------------ seam B2
    def distances(self, queries, want_counts=True):
        """(counts uint32[Q, n_rows, 2] or None, dist float64[Q, n_rows]) in caller row order."""
        q = self._queries(queries)
        dist = np.empty((len(q), self.n_rows), np.float64)
        counts = np.empty((len(q), self.n_rows, 2), np.uint32) if want_counts else None
        self._check(self.lib.apples_distances(self.ctx, _ptr(q), len(q), _ptr(counts), _ptr(dist)))
        return counts, dist

    # ------------------------------------------------------------------ seam B1
    def place_sequences(self, queries, self_rows=None):
        q = self._queries(queries)
        out = np.zeros(len(q), PLACEMENT_DTYPE)
        sr = self._self_rows(self_rows, len(q))
        self._check(self.lib.apples_place_from_sequences(self.ctx, _ptr(q), len(q), _ptr(sr), _ptr(out)))
        return out

    def place_sequences_streamed(self, queries, self_rows=None):
        """Same work, placements left on the device: returns (handle, n) for fetch / placements_device_ptr
        / free_queries."""
        q = self._queries(queries)
        sr = self._self_rows(self_rows, len(q))
        h = C.c_int64()
        self._check(self.lib.apples_place_sequences_streamed(self.ctx, _ptr(q), len(q), _ptr(sr), C.byref(h)))
        return h.value, len(q)

    @staticmethod
    def _self_rows(self_rows, n):
        if self_rows is None:
            return None
        sr = np.ascontiguousarray(self_rows, np.int32)
        if sr.shape != (n,):
            raise ValueError('self_rows must have one entry per query')
        return sr

    def place_distances(self, dist, col_nodes, self_cols=None):
        d = np.ascontiguousarray(dist, np.float64)
        if d.ndim == 1:
            d = d.reshape(1, -1)
        cn = np.ascontiguousarray(col_nodes, np.int32)
        assert d.shape[1] == len(cn)
        out = np.zeros(len(d), PLACEMENT_DTYPE)
        sc = np.ascontiguousarray(self_cols, np.int32) if self_cols is not None else None
        self._check(self.lib.apples_place_from_distances(self.ctx, _ptr(d), d.shape[0], d.shape[1], _ptr(cn), _ptr(sc),
                                                         _ptr(out)))
        return out

    # ------------------------------------------------------------------ seam B3 (inspection)
    def sweep_edges(self, obs_nodes, obs_dist):
        n = self.tree.n_nodes
        on = np.ascontiguousarray(obs_nodes, np.int32)
        od = np.ascontiguousarray(obs_dist, np.float64)
        valid = np.zeros(n, np.uint8)
        S = np.zeros((n, 6)); R = np.zeros((n, 6)); x = np.zeros((n, 4)); err = np.zeros(n)
        lca = np.zeros(1, np.int32)
        out = np.zeros(1, PLACEMENT_DTYPE)
        self._check(self.lib.apples_sweep_edges(self.ctx, _ptr(on), _ptr(od), len(on), _ptr(valid), _ptr(S), _ptr(R),
                                                _ptr(x), _ptr(err), _ptr(lca), _ptr(out)))
        return dict(valid=valid.astype(bool), S=S, R=R, x=x, err=err, lca=int(lca[0]), placement=out[0])

    # ------------------------------------------------------------------ resident blocks (bench)
    def upload_queries(self, queries, self_rows=None):
        q = self._queries(queries)
        sr = self._self_rows(self_rows, len(q))
        h = C.c_int64()
        self._check(self.lib.apples_queries_upload(self.ctx, _ptr(q), len(q), _ptr(sr), C.byref(h)))
        return h.value, len(q)

    def upload_table(self, dist, col_nodes, self_cols=None):
        d = np.ascontiguousarray(dist, np.float64)
        cn = np.ascontiguousarray(col_nodes, np.int32)
        sc = np.ascontiguousarray(self_cols, np.int32) if self_cols is not None else None
        h = C.c_int64()
        self._check(self.lib.apples_table_upload(self.ctx, _ptr(d), d.shape[0], d.shape[1], _ptr(cn), _ptr(sc), C.byref(h)))
        return h.value, d.shape[0]

    def free_queries(self, handle):
        self._check(self.lib.apples_queries_free(self.ctx, handle))

    def place_resident(self, handle):
        self._check(self.lib.apples_place_resident(self.ctx, handle))

    def fetch(self, handle, n):
        out = np.zeros(n, PLACEMENT_DTYPE)
        self._check(self.lib.apples_fetch_placements(self.ctx, handle, _ptr(out)))
        return out

    def placements_device_ptr(self, handle):
        p = C.c_void_p()
        self._check(self.lib.apples_placements_device_ptr(self.ctx, handle, C.byref(p)))
        return p.value

    def distances_resident(self, handle, query_tile=0):
        self._check(self.lib.apples_distances_resident(self.ctx, handle, int(query_tile)))


def placement_rows(out):
    """Vectorised :func:`placement_row`: list of p rows for a placement array."""
    flags = out['flags']
    trivial = (flags & (F_EXACT | F_INSUFFICIENT | F_DEGENERATE)) != 0
    pint = (flags & F_PENDANT_INT) != 0
    edge = out['edge'].tolist()
    err = out['error'].tolist()
    dist = out['distal'].tolist()
    pend = out['pendant'].tolist()
    return [[e, 0, 1, 0, 0] if t else [e, er, 1, d, 0 if pi else pe]
            for e, er, d, pe, t, pi in zip(edge, err, dist, pend, trivial.tolist(), pint.tolist())]


def placement_row(p):
    """apples_placement -> the jplace p row with the reference's int/float leakage."""
    flags = int(p['flags'])
    if flags & (F_EXACT | F_INSUFFICIENT | F_DEGENERATE):
        return [int(p['edge']), 0, 1, 0, 0]  # PoolQueryWorker.py:37,74,88
    pend = 0 if flags & F_PENDANT_INT else float(p['pendant'])
    return [int(p['edge']), float(p['error']), 1, float(p['distal']), pend]
