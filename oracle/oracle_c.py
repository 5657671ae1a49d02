"""ctypes binding of the C oracle (oracle/oracle.c).  TEST INFRASTRUCTURE ONLY -- see the header
of oracle.c.  Same call shapes as apples_amd.engine.Engine so parity tests read side by side."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, '_build', 'liboracle.so')

PLACEMENT_DTYPE = np.dtype([('edge', '<i4'), ('flags', '<u4'), ('error', '<f8'), ('distal', '<f8'),
                            ('pendant', '<f8'), ('n_obs', '<i4'), ('n_valid', '<i4')], align=True)
METHODS = {'OLS': 0, 'FM': 1, 'BME': 2, 'BE': 3}
CRITERIA = {'MLSE': 0, 'ME': 1, 'HYBRID': 2}


class _Tree(C.Structure):
    _fields_ = [('n_nodes', C.c_int32), ('parent', C.c_void_p), ('edge_len', C.c_void_p), ('child_off', C.c_void_p),
                ('child_idx', C.c_void_p), ('level', C.c_void_p)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(_HERE, 'oracle.c')):
            subprocess.check_call(['make', '-s', '-C', _HERE])
        _lib = C.CDLL(LIB)
        _lib.orc_jc69_from_counts.restype = C.c_double
        _lib.orc_jc69_from_counts.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_double, C.c_void_p]
        _lib.orc_scoredist.restype = C.c_double
        _lib.orc_scoredist.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p]
        _lib.orc_select.restype = C.c_int64
    return _lib


class COracle:
    def __init__(self, tree, ref_seqs=None, ref_nodes=None, clusters=None, protein=False, method='FM', criterion='MLSE',
                 negative=False, threshold=0.2, baseobs=25, overlap=0.001, lut=None, threads=1):
        self.lib = load()
        self.tree = tree
        self._keep = [np.ascontiguousarray(tree.parent, np.int32), np.ascontiguousarray(tree.edge_len, np.float64),
                      np.ascontiguousarray(tree.child_off, np.int32), np.ascontiguousarray(tree.child_idx, np.int32),
                      np.ascontiguousarray(tree.level, np.int32)]
        t = _Tree()
        t.n_nodes = tree.n_nodes
        t.parent, t.edge_len, t.child_off, t.child_idx, t.level = [_p(a) for a in self._keep]
        self.t = t
        self.protein = protein
        self.opts = dict(method=method, criterion=criterion, negative=negative, threshold=threshold, baseobs=baseobs,
                         overlap=overlap)
        self.threads = threads
        self.lut = np.ascontiguousarray(lut, np.float64) if lut is not None else None
        from apples_oracle import BLOSUM45
        self.blosum = np.ascontiguousarray(BLOSUM45, np.float64)
        self.rows = None
        if ref_seqs is not None:
            ref_seqs = np.ascontiguousarray(ref_seqs, np.uint8)
            self.n_refs, self.L = ref_seqs.shape
            self.rep_row = self.member_off = self.member_row = None
            rows = ref_seqs
            self.n_reps = self.n_refs
            if clusters is not None:
                cons, rep_row, member_off, member_row = clusters
                cons = np.ascontiguousarray(cons, np.uint8).reshape(-1, self.L)
                rows = np.ascontiguousarray(np.vstack([ref_seqs, cons])) if len(cons) else ref_seqs
                self.rep_row = np.ascontiguousarray(rep_row, np.int32)
                self.member_off = np.ascontiguousarray(member_off, np.int32)
                self.member_row = np.ascontiguousarray(member_row, np.int32)
                self.n_reps = len(self.rep_row)
            self.rows = rows
            self.ref_nodes = np.ascontiguousarray(ref_nodes, np.int32)

    def set_options(self, **kw):
        self.opts.update(kw)

    def _mc(self):
        o = self.opts
        return METHODS.get(o['method'], 0), CRITERIA.get(o['criterion'], 0), 1 if o['negative'] else 0

    def distances(self, queries):
        q = np.ascontiguousarray(queries, np.uint8).reshape(-1, self.L)
        out = np.empty((len(q), len(self.rows)))
        for i in range(len(q)):
            self.lib.orc_distance_row(_p(q[i]), _p(self.rows), C.c_int64(len(self.rows)), C.c_int(self.L),
                                      C.c_int(1 if self.protein else 0), C.c_double(self.opts['overlap']), _p(self.lut),
                                      _p(self.blosum), _p(out[i]))
        return out

    def place_sequences(self, queries, self_rows=None):
        q = np.ascontiguousarray(queries, np.uint8).reshape(-1, self.L)
        out = np.zeros(len(q), PLACEMENT_DTYPE)
        sr = np.ascontiguousarray(self_rows, np.int32) if self_rows is not None else None
        m, c, n = self._mc()
        o = self.opts
        self.lib.orc_place_from_sequences(C.byref(self.t), _p(self.rows), C.c_int64(len(self.rows)), C.c_int64(self.n_refs),
                                          C.c_int(self.L), _p(self.ref_nodes), C.c_int64(self.n_reps), _p(self.rep_row),
                                          _p(self.member_off), _p(self.member_row), C.c_int(1 if self.protein else 0),
                                          C.c_int(m), C.c_int(c), C.c_int(n), C.c_double(o['threshold']),
                                          C.c_int(o['baseobs']), C.c_double(o['overlap']), _p(self.lut), _p(self.blosum),
                                          _p(q), C.c_int64(len(q)), _p(sr), _p(out), C.c_int(self.threads))
        return out

    def place_distances(self, dist, col_nodes, self_cols=None):
        d = np.ascontiguousarray(dist, np.float64)
        if d.ndim == 1:
            d = d.reshape(1, -1)
        cn = np.ascontiguousarray(col_nodes, np.int32)
        sc = np.ascontiguousarray(self_cols, np.int32) if self_cols is not None else None
        out = np.zeros(len(d), PLACEMENT_DTYPE)
        m, c, n = self._mc()
        o = self.opts
        self.lib.orc_place_from_distances(C.byref(self.t), _p(d), C.c_int64(d.shape[0]), C.c_int64(d.shape[1]), _p(cn),
                                          _p(sc), C.c_int(m), C.c_int(c), C.c_int(n), C.c_double(o['threshold']),
                                          C.c_int(o['baseobs']), _p(out), C.c_int(self.threads))
        return out

    def sweep_edges(self, obs_nodes, obs_dist):
        n = self.tree.n_nodes
        on = np.ascontiguousarray(obs_nodes, np.int32)
        od = np.ascontiguousarray(obs_dist, np.float64)
        valid = np.zeros(n, np.uint8)
        S = np.zeros((n, 6)); R = np.zeros((n, 6)); x = np.zeros((n, 4)); err = np.zeros(n)
        lca = np.zeros(1, np.int32)
        out = np.zeros(1, PLACEMENT_DTYPE)
        m, c, ng = self._mc()
        self.lib.orc_sweep_edges(C.byref(self.t), _p(on), _p(od), C.c_int(len(on)), C.c_int(m), C.c_int(c), C.c_int(ng),
                                 _p(valid), _p(S), _p(R), _p(x), _p(err), _p(lca), _p(out))
        return dict(valid=valid.astype(bool), S=S, R=R, x=x, err=err, lca=int(lca[0]), placement=out[0])


def libm_log_array(x):
    """libm's log of every element (the host routine the C oracle's distances go through)."""
    lib = load()
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    lib.orc_log_array(_p(x), _p(out), C.c_int64(x.size))
    return out
