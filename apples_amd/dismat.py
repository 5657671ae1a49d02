"""Distance-table input (run_apples.py -d): the reference's text format (run_apples.py:43-54), read
natively for tables of 10^8 and more values, and a binary ``.npz`` form."""
import ctypes
import mmap
import re

import numpy as np


def _dedupe(tags, D):
    """dict(zip(tags, values)) semantics: a repeated tag keeps its first position and its last value."""
    if len(set(tags)) == len(tags):
        return list(tags), D
    keep, seen = [], {}
    for i, c in enumerate(tags):
        if c in seen:
            D[:, seen[c]] = D[:, i]
        else:
            seen[c] = len(keep)
            keep.append(i)
            if len(keep) - 1 != i:
                D[:, len(keep) - 1] = D[:, i]
    return [tags[i] for i in keep], np.ascontiguousarray(D[:, :len(keep)])


def read_dismat_py(f):
    """Header: whitespace-split names after the first field; rows: name then floats
    (run_apples.py:43-54).  Returns (query names, column names, float64 matrix); -1 = no value."""
    tags = re.split(r'\s+', f.readline().rstrip())[1:]
    cols, seen = [], {}
    for t in tags:  # dict(zip(tags, ...)): a repeated name keeps its first position, last value
        if t not in seen:
            seen[t] = len(cols)
            cols.append(t)
    names, rows = [], []
    for line in f.readlines():
        d = re.split(r'\s+', line.strip())
        names.append(d[0])
        row = np.full(len(cols), -1.0)
        for t, v in zip(tags, d[1:]):
            row[seen[t]] = float(v)
        rows.append(row)
    return names, cols, (np.vstack(rows) if rows else np.zeros((0, len(cols))))


def read_dismat_text(path):
    """The same through libapples_io.so's scanner (include/apples_io.h: apples_dismat_scan) on the mapped
    file image; anything it does not take (no library, values that are not plain decimals, non-ASCII
    names) goes to :func:`read_dismat_py`."""
    from .fasta import _load_io
    lib = _load_io()
    if lib is not None and hasattr(lib, 'apples_dismat_scan'):
        with open(path, 'rb') as fh:
            size = fh.seek(0, 2)
            if size > 0:
                mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
                try:
                    img = np.frombuffer(mm, dtype=np.uint8)
                    got = _scan(lib, img)
                    del img
                finally:
                    try:
                        mm.close()
                    except BufferError:
                        pass
                if got is not None:
                    return got
    with open(path) as f:
        return read_dismat_py(f)


def _scan(lib, img):
    P = ctypes.c_void_p
    fn = lib.apples_dismat_scan
    fn.restype = ctypes.c_int
    fn.argtypes = [P, ctypes.c_int64, P, ctypes.c_int64, ctypes.c_int64, P, P, P, P, P, P]
    nt, nr = ctypes.c_int64(), ctypes.c_int64()
    ptr = img.ctypes.data_as(P)
    if fn(ptr, img.size, None, 0, 0, ctypes.byref(nt), ctypes.byref(nr), None, None, None, None) != 0:
        return None
    nt, nr = nt.value, nr.value
    D = np.empty((nr, nt), np.float64)
    tag_off, tag_len = np.zeros(max(nt, 1), np.int64), np.zeros(max(nt, 1), np.int32)
    name_off, name_len = np.zeros(max(nr, 1), np.int64), np.zeros(max(nr, 1), np.int32)
    a, b = ctypes.c_int64(), ctypes.c_int64()
    rc = fn(ptr, img.size, D.ctypes.data_as(P), nt, nr, ctypes.byref(a), ctypes.byref(b), tag_off.ctypes.data_as(P),
            tag_len.ctypes.data_as(P), name_off.ctypes.data_as(P), name_len.ctypes.data_as(P))
    if rc != 0:
        return None
    raw = img.tobytes() if img.size < (1 << 20) else None

    def text(off, ln):
        return (raw[off:off + ln] if raw is not None else img[off:off + ln].tobytes()).decode('ascii')
    try:
        tags = [text(int(tag_off[i]), int(tag_len[i])) for i in range(nt)]
        names = [text(int(name_off[i]), int(name_len[i])) for i in range(nr)]
    except UnicodeDecodeError:
        return None
    cols, D = _dedupe(tags, D)
    return names, cols, D


def read_dismat_binary(path):
    """Binary form of the same table (a 200 k-column table is 20 GB of text): a numpy ``.npz`` with
    ``queries`` and ``columns`` (string arrays) and ``D`` (float64 [queries, columns], negative =
    missing).  Written by ``numpy.savez(path, queries=..., columns=..., D=...)``."""
    z = np.load(path, allow_pickle=False)
    names = [str(x) for x in z['queries']]
    cols = [str(x) for x in z['columns']]
    D = np.ascontiguousarray(z['D'], dtype=np.float64)
    if D.shape != (len(names), len(cols)):
        raise ValueError('distance table shape %s does not match %d queries x %d columns' % (D.shape, len(names), len(cols)))
    cols, D = _dedupe(cols, D)
    return names, cols, D


def read_dismat(path):
    with open(path, 'rb') as f:
        magic = f.read(2)
    return read_dismat_binary(path) if magic == b'PK' else read_dismat_text(path)
