"""jplace assembly and emission (apples/jutil.py:1-19, run_apples.py:106-118)."""
import json


def join_jplace(lst):
    """Concatenate per-query results, dropping unplaceable ones (edge -1) -- except that the first
    result is kept as is whenever there is more than one (apples/jutil.py:11-18)."""
    result = lst[0]
    if len(lst) == 1:
        if result['placements'][0]['p'][0][0] == -1:
            result['placements'] = []
    else:
        extra = [r['placements'][0] for r in lst[1:] if r['placements'][0]['p'][0][0] != -1]
        result['placements'] = result['placements'] + extra
    return result


def finish(result, tree_string, argv):
    result['tree'] = tree_string
    result['metadata'] = {'invocation': ' '.join(argv)}
    result['fields'] = ['edge_num', 'likelihood', 'like_weight_ratio', 'distal_length', 'pendant_length']
    result['version'] = 3
    return result


def dumps(result):
    return json.dumps(result, sort_keys=True, indent=4) + '\n'


# ---------------------------------------------------------------------------------------------
# Streaming writer.  json.dumps with indent runs the pure-Python encoder over one
# giant dict: 1.4 s per 100 000 placements, several times the whole GPU pass.  The text below is
# the same text, byte for byte, assembled directly.

def _num(x):
    if isinstance(x, float):
        r = float.__repr__(x)
        if r in ('nan', 'inf', '-inf'):  # json's allow_nan spellings
            return {'nan': 'NaN', 'inf': 'Infinity', '-inf': '-Infinity'}[r]
        return r
    return str(int(x))


def iter_text(placements, tree_string, argv):
    """Yield the jplace text in pieces; ``placements`` is an iterable of (name, p row), already
    filtered as :func:`join_jplace` would (see :func:`keep_mask`).  Same bytes as
    ``dumps(finish(join_jplace(results), tree_string, argv))``."""
    q = json.dumps
    yield ('{\n    "fields": [\n        "edge_num",\n        "likelihood",\n        "like_weight_ratio",\n'
           '        "distal_length",\n        "pendant_length"\n    ],\n    "metadata": {\n        "invocation": %s\n    },\n'
           % q(' '.join(argv)))
    first = True
    buf = []
    for name, row in placements:
        buf.append('%s\n        {\n            "n": [\n                %s\n            ],\n            "p": [\n'
                   '                [\n                    %s,\n                    %s,\n                    %s,\n'
                   '                    %s,\n                    %s\n                ]\n            ]\n        }'
                   % ('    "placements": [' if first else ',', q(name), _num(row[0]), _num(row[1]), _num(row[2]),
                      _num(row[3]), _num(row[4])))
        first = False
        if len(buf) >= 4096:
            yield ''.join(buf)
            buf = []
    if buf:
        yield ''.join(buf)
    yield '    "placements": [],\n' if first else '\n    ],\n'
    yield '    "tree": %s,\n    "version": 3\n}\n' % q(tree_string)


def keep_mask(edges):
    """Which results survive :func:`join_jplace`: unplaceable ones (edge -1) are dropped, except
    that the first result is kept as is whenever there is more than one (apples/jutil.py:11-18)."""
    keep = [e != -1 for e in edges]
    if len(keep) > 1:
        keep[0] = True
    return keep


def write_native(f, names, cols, tree_string, argv, chunk=16384):
    """The same text as :func:`iter_text` written to the binary file ``f``, the placement rows formatted by libapples_io.so
    (apples_jplace_rows: 100 000 rows in 0.03 s where the Python loop above takes 0.3 s).  ``cols`` = the p rows as columns
    (worker._rows(arrays=True)); rows are filtered as :func:`join_jplace` would.  Returns False -- nothing written -- when the
    library is missing or a name holds a character json.dumps would escape (the caller then uses :func:`iter_text`)."""
    import ctypes
    import numpy as np
    from .fasta import _load_io
    lib = _load_io()
    if lib is None or not hasattr(lib, 'apples_jplace_rows'):
        return False
    n = len(names)
    blob = '\n'.join(names).encode('utf-8', 'surrogatepass') if n else b''
    b = np.frombuffer(blob, dtype=np.uint8)
    plain = (b >= 0x20) & (b <= 0x7e) & (b != 0x22) & (b != 0x5c)
    if n and int(plain.sum()) != len(b) - (n - 1):  # everything but the separators must be printable ASCII without " and \
        return False
    lens = np.fromiter((len(x) for x in names), dtype=np.int32, count=n)
    if n and int(lens.sum()) + n - 1 != len(b):
        return False
    offs = np.zeros(n, dtype=np.int64)
    if n:
        np.cumsum(lens[:-1].astype(np.int64) + 1, out=offs[1:])
    edge = np.ascontiguousarray(cols['edge'], dtype=np.int32)
    keep = (edge != -1).astype(np.uint8)
    if n > 1:
        keep[0] = 1  # the first result is kept as it is whenever there is more than one (apples/jutil.py:11-18)
    err = np.ascontiguousarray(cols['error'], dtype=np.float64)
    dist = np.ascontiguousarray(cols['distal'], dtype=np.float64)
    pend = np.ascontiguousarray(cols['pendant'], dtype=np.float64)
    kind = np.ascontiguousarray(cols['kind'], dtype=np.uint8)
    lib.apples_jplace_rows.restype = ctypes.c_int64
    lib.apples_jplace_rows.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] + [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_void_p,
                                                                                                        ctypes.c_int64]
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    q = json.dumps
    f.write(('{\n    "fields": [\n        "edge_num",\n        "likelihood",\n        "like_weight_ratio",\n'
             '        "distal_length",\n        "pendant_length"\n    ],\n    "metadata": {\n        "invocation": %s\n    },\n'
             % q(' '.join(argv))).encode())
    first = True
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        if not keep[lo:hi].any():
            continue
        cap = 512 * (hi - lo) + int(lens[lo:hi].sum())
        out = np.empty(cap, dtype=np.uint8)
        k = lib.apples_jplace_rows(ptr(b) if n else None, ptr(offs[lo:]), ptr(lens[lo:]), hi - lo, ptr(edge[lo:]), ptr(err[lo:]),
                                   ptr(dist[lo:]), ptr(pend[lo:]), ptr(kind[lo:]), ptr(keep[lo:]), 1 if first else 0, ptr(out), cap)
        if k < 0:
            raise RuntimeError('apples_jplace_rows: output buffer too small')
        f.write(out[:k].tobytes())
        first = False
    f.write(b'    "placements": [],\n' if first else b'\n    ],\n')
    f.write(('    "tree": %s,\n    "version": 3\n}\n' % q(tree_string)).encode())
    return True
