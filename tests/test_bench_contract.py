"""bench.py's bookkeeping without a GPU: the workload table against BASELINE.json's configs and the roofline arithmetic
(SURVEY 8d's algorithmic bytes, DESIGN section 4's operation counts) on made-up timings."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _placements(n, swept):
    p = np.zeros(n, dtype=[('edge', '<i4'), ('flags', '<u4'), ('error', '<f8'), ('distal', '<f8'), ('pendant', '<f8'),
                           ('n_obs', '<i4'), ('n_valid', '<i4')])
    p['n_valid'] = swept - 1  # roofline_of counts n_valid + 1 swept nodes per placed query
    return p


def test_workloads_are_the_baseline_configs():
    cfg = json.load(open(os.path.join(ROOT, 'BASELINE.json')))['configs']
    w = bench.WORKLOADS
    assert w['c2'][:3] == (10000, 1000, 10000) and '10k-leaf' in cfg[1] and '10k queries' in cfg[1]
    assert w['c3'][:3] == (200000, 1000, 100000) and '200k-leaf' in cfg[2] and '100k queries' in cfg[2]
    assert w['c4'][:4] == (50000, 500, 50000, True) and w['c4'][4] == 'FM' and 'scoredist+FM' in cfg[3]
    assert w['c5'][0] == 200000 and w['c5'][4] == 'BME' and 'BME' in cfg[4]
    assert all(v[5] == 0.2 for k, v in w.items() if not k.endswith('-all'))  # -f 0.2, the stress variants -f 1e9


def test_roofline_of_prices_the_matrix_core_distance_pass_against_the_fp4_peak():
    nq, rows, L = 100000, 200000, 1000
    per = {'dist_ms': 27.4, 'select_ms': 4.1, 'sweep_ms': 15.3}
    r = bench.roofline_of('c3', nq, rows, L, False, False, per, 5, _placements(nq, 2400), {'code_planes': 2, 'all_singleton': 1})
    ops = 2.0 * 4.0 * nq * rows * 1024  # four components per site, sites padded to whole 32-site words
    assert r['kernel'] == 'jc69_distance' and r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 10000.0
    assert abs(r['achieved'] - ops / 27.4e-3 / 1e12) < 1e-6 and abs(r['frac'] - r['achieved'] / 10000.0) < 1e-12
    assert r['kernel_calls_per_step'] == 5 and abs(r['dominant_kernel_ms_per_step'] - 27.4) < 1e-12 and r['device_batches'] >= 1
    assert abs(r['sweep_hbm']['algorithmic_GBps'] - 332.0 * 2400 * nq / 15.3e-3 / 1e9) < 1e-3
    # SURVEY 8d's byte figures ride along: N (L + 8) + L per query, 332 V per query
    assert abs(r['hbm_algorithmic_GBps'] - nq * (rows * (L + 8.0) + L) / 27.4e-3 / 1e9) < 1e-3
    assert abs(r['all_kernels_GBps']['lsq_sweep'] - 332.0 * 2400 * nq / 15.3e-3 / 1e9) < 1e-3


def test_roofline_of_table_input_and_sweep_bound_workloads():
    nq, cols = 12500, 200000
    per = {'dist_ms': 0.0, 'select_ms': 4.0, 'sweep_ms': 2.4}
    r = bench.roofline_of('c5', nq, cols, 0, False, True, per, 0, _placements(nq, 2000), {})
    assert r['kernel'] == 'table_select' and r['bound'] == 'hbm' and r['peak'] == 8000.0
    assert abs(r['achieved'] - nq * cols * 8.0 / 4.0e-3 / 1e9) < 1e-6 and abs(r['frac'] - r['achieved'] / 8000.0) < 1e-12
    # every leaf observed: the sweep dominates and is priced by 332 V bytes per query
    per = {'dist_ms': 0.9, 'select_ms': 3.4, 'sweep_ms': 51.0}
    r = bench.roofline_of('c3-all', 2048, 200000, 1000, False, False, per, 1, _placements(2048, 399999), {'code_planes': 2, 'all_singleton': 1})
    assert r['kernel'] == 'lsq_sweep' and r['bound'] == 'hbm'
    assert abs(r['achieved'] - 332.0 * 399999 * 2048 / 51.0e-3 / 1e9) < 1e-3


def test_committed_counter_summary_names_every_workload_and_one_commit():
    d = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_summary.json')))
    assert set(d) >= {'c2', 'c3', 'c4', 'c5', 'c3-clustered'}
    assert len({d[w].get('measured_at_commit') for w in d}) == 1
    bytes_, commit = bench.load_traffic('c3', 'jc69_distance')
    assert bytes_ and bytes_ > 1e9 and commit == d['c3']['measured_at_commit']
