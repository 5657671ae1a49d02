"""Shared helpers for the parity tests."""
import json
import math
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
DATA = os.path.join(GOLD, 'data')


def load_json(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def read_dismat(path):
    """(query name, {column name: float}) rows of a distance table (run_apples.py:43-54)."""
    with open(path) as f:
        tags = re.split(r'\s+', f.readline().rstrip())[1:]
        for line in f.readlines():
            d = re.split(r'\s+', line.strip())
            yield d[0], dict(zip(tags, map(float, d[1:])))


def close(a, b, rel=1e-9, abs_tol=1e-12):
    return a == b or math.isclose(a, b, rel_tol=rel, abs_tol=abs_tol)


def assert_prow(got, want, rel=1e-6, err_abs=1e-9, ctx=''):
    """p rows [edge, err, 1, distal, pendant]: edge exact, lengths rel 1e-6 (north_star), error
    compared with an absolute floor (it is a cancelling sum that may legitimately be ~1e-16)."""
    assert got[0] == want[0], '%s edge %r != %r (got %r want %r)' % (ctx, got[0], want[0], got, want)
    assert got[2] == want[2]
    assert math.isclose(got[1], want[1], rel_tol=rel, abs_tol=err_abs), '%s err %r vs %r' % (ctx, got[1], want[1])
    assert math.isclose(got[3], want[3], rel_tol=rel, abs_tol=1e-12), '%s distal %r vs %r' % (ctx, got[3], want[3])
    assert math.isclose(got[4], want[4], rel_tol=rel, abs_tol=1e-12), '%s pendant %r vs %r' % (ctx, got[4], want[4])
    # int-vs-float leakage into jplace: clamped pendant is the int 0
    assert isinstance(got[4], int) == isinstance(want[4], int), '%s pendant type %r vs %r' % (ctx, got[4], want[4])
    assert isinstance(got[1], int) == isinstance(want[1], int), '%s err type %r vs %r' % (ctx, got[1], want[1])
    assert isinstance(got[3], int) == isinstance(want[3], int), '%s distal type %r vs %r' % (ctx, got[3], want[3])


def prow_is_tie(got, want, rel=1e-9):
    """The tie class of SURVEY H1: edges that meet at a node have mathematically equal residuals there, so which of them wins is
    decided by the last bit of the distances -- and scoredist's are reproducible only to 1e-15 (BLAS summation order, device log).
    True when the two rows name different edges but the same residual and the same pendant length."""
    return (got[0] != want[0] and got[0] >= 0 and want[0] >= 0 and
            math.isclose(got[1], want[1], rel_tol=rel, abs_tol=1e-15) and math.isclose(got[4], want[4], rel_tol=rel, abs_tol=1e-12))
