"""FASTA/FASTQ reading and the byte encoding the distance kernels see.

Behaviour follows apples/fasta2dic.py:
* record name = header up to the first space (fasta2dic.py:17);
* the last character of every line is dropped unconditionally (``l[:-1]``,
  fasta2dic.py:13,20,22), so a final line without a newline loses a base;
* sequences are upper-cased, or with masking lower-case -> '-' (fasta2dic.py:50,63-67);
* letters outside the alphabet -> '-': nt ``BDEFHIJKLMNOPQRSUVWXYZ``, aa ``BJOUXZ``
  (fasta2dic.py:58-61); every other byte passes through as an ordinary symbol.

Instead of ``{name: ndarray('S1')}`` the build keeps one dense ``uint8[N, L]``
matrix plus a name list (dict semantics: a repeated name keeps its first
position and the last sequence).
"""
import os

import numpy as np


def read_records(fp):
    """Yield ``(name, seq)`` for FASTA records and ``(name, seq)`` for FASTQ ones."""
    last = None
    while True:
        if not last:
            for line in fp:
                if line[0] in '>@':
                    last = line[:-1]
                    break
        if not last:
            break
        name = last[1:].partition(' ')[0]
        parts = []
        last = None
        for line in fp:
            if line[0] in '@+>':
                last = line[:-1]
                break
            parts.append(line[:-1])
        if not last or last[0] != '+':
            yield name, ''.join(parts)
            if not last:
                break
        else:  # FASTQ: skip the quality block
            seq = ''.join(parts)
            got = 0
            complete = False
            for line in fp:
                got += len(line) - 1
                if got >= len(seq):
                    last = None
                    complete = True
                    yield name, seq
                    break
            if not complete:
                yield name, seq
                break


def _translation(prot_flag, mask_flag):
    tab = np.arange(256, dtype=np.uint8)
    lower = np.frombuffer(b'abcdefghijklmnopqrstuvwxyz', dtype=np.uint8)
    if mask_flag:
        tab[lower] = ord('-')
    else:
        tab[lower] = lower - 32
    invalid = b'BJOUXZ' if prot_flag else b'BDEFHIJKLMNOPQRSUVWXYZ'
    inv = np.zeros(256, dtype=bool)
    inv[np.frombuffer(invalid, dtype=np.uint8)] = True
    tab[inv[tab]] = ord('-')
    return tab


def encode_sequence(seq, prot_flag, mask_flag):
    """str -> uint8 array in the reference's byte encoding."""
    raw = np.frombuffer(seq.encode(), dtype=np.uint8)
    return _translation(prot_flag, mask_flag)[raw]


class Alignment:
    """Names + dense ``uint8[N, L]`` matrix (row r is ``names[r]``)."""

    def __init__(self, names, seqs):
        self.names = list(names)
        self.seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        self.index = {n: i for i, n in enumerate(self.names)}

    def __len__(self):
        return len(self.names)

    @property
    def length(self):
        return self.seqs.shape[1] if self.seqs.ndim == 2 else 0


def _read_alignment_py(path, prot_flag, mask_flag):
    """Record-by-record reader (the restatement the native scanner is tested against)."""
    tab = _translation(prot_flag, mask_flag)
    order = {}
    rows = []
    with open(path) as f:
        for name, seq in read_records(f):
            enc = tab[np.frombuffer(seq.encode(), dtype=np.uint8)]
            if name in order:
                rows[order[name]] = enc
            else:
                order[name] = len(rows)
                rows.append(enc)
    if not rows:
        return Alignment([], np.zeros((0, 0), dtype=np.uint8))
    L = len(rows[0])
    for n, r in zip(order, rows):
        if len(r) != L:
            raise ValueError('sequence %s has length %d, expected %d (alignment rows must be equal length)'
                             % (n, len(r), L))
    return Alignment(list(order), np.vstack(rows))


_io_lib = None


def _load_io():
    """libapples_io.so (include/apples_io.h), or None when it has not been built."""
    global _io_lib
    if _io_lib is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libapples_io.so')
        if not os.path.exists(path):
            _io_lib = False
        else:
            lib = ctypes.CDLL(path)
            lib.apples_fasta_scan.restype = ctypes.c_int
            lib.apples_fasta_scan.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_int64] + [ctypes.c_void_p] * 6
            if hasattr(lib, 'apples_fasta_scan_mt'):
                lib.apples_fasta_scan_mt.restype = ctypes.c_int
                lib.apples_fasta_scan_mt.argtypes = lib.apples_fasta_scan.argtypes + [ctypes.c_int32]
            if hasattr(lib, 'apples_newick_scan'):
                lib.apples_newick_scan.restype = ctypes.c_int
                lib.apples_newick_scan.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64] + \
                    [ctypes.c_void_p] * 11
            if hasattr(lib, 'apples_max_clusters'):
                lib.apples_max_clusters.restype = ctypes.c_int
                lib.apples_max_clusters.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_int32, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p]
            _io_lib = lib
    return _io_lib or None


def read_alignment(path, prot_flag, mask_flag):
    """FASTA file -> :class:`Alignment` (apples/fasta2dic.py:42-72).  With libapples_io.so the file
    image is scanned natively, every sequence written straight into its row of the dense matrix
    (SURVEY 8f-2); repeated names, ragged input or a missing library take the record-by-record path."""
    lib = _load_io()
    if lib is None:
        return _read_alignment_py(path, prot_flag, mask_flag)
    import ctypes
    tab = _translation(prot_flag, mask_flag)
    with open(path, 'rb') as f:
        data = np.frombuffer(f.read(), dtype=np.uint8)
    n_rec = ctypes.c_int64(0)
    length = ctypes.c_int64(0)
    bad = ctypes.c_int64(-1)
    bad_len = ctypes.c_int64(0)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    threads = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else 0

    def scan(rows, n_rows, off, ln):
        # plain FASTA with "\n" line ends: the threaded scanner; anything else (FASTQ, "\r", ...): the general one
        rc = 3
        if hasattr(lib, 'apples_fasta_scan_mt'):
            rc = lib.apples_fasta_scan_mt(ptr(data), data.size, ptr(tab), rows, n_rows, ctypes.byref(n_rec), ctypes.byref(length),
                                          off, ln, ctypes.byref(bad), ctypes.byref(bad_len), min(threads, 32))
        if rc == 3:
            rc = lib.apples_fasta_scan(ptr(data), data.size, ptr(tab), rows, n_rows, ctypes.byref(n_rec), ctypes.byref(length),
                                       off, ln, ctypes.byref(bad), ctypes.byref(bad_len))
        return rc

    scan(None, 0, None, None)
    n, L = n_rec.value, length.value
    if n == 0:
        return Alignment([], np.zeros((0, 0), dtype=np.uint8))
    mat = np.empty((n, L), dtype=np.uint8)
    off = np.empty(n, dtype=np.int64)
    ln = np.empty(n, dtype=np.int32)
    rc = scan(ptr(mat), n, ptr(off), ptr(ln))
    raw = data.tobytes() if n < 64 else memoryview(data)
    names = [bytes(raw[o:o + k]).decode() for o, k in zip(off.tolist(), ln.tolist())]
    if rc != 0 or len(set(names)) != n:
        return _read_alignment_py(path, prot_flag, mask_flag)  # dict semantics / the reference's error
    return Alignment(names, mat)
