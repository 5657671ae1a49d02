#!/usr/bin/env python3
"""profiles/pmc_summary.json from a scripts/pmc_passes.sh run: HBM bytes per launch per kernel.

FETCH_SIZE and WRITE_SIZE are in KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half
of the bytes of a wide coalesced streaming read (16 B/lane) -> doubled for kernels that stream
(distance, selection); other access widths are uncalibrated -> the sweep's scattered 64-byte record
traffic is reported raw, with the doubled figure alongside.

usage: pmc_to_traffic.py <gpurun_out/pmc_dir> <workload> [<out json>]"""
import json, os, subprocess, sys
d, workload = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'pmc_summary.json')
summ = json.loads(subprocess.check_output([sys.executable, os.path.join(os.path.dirname(__file__), 'pmc_summary.py'), d]))
names = {'k_jc69_gemm': 'jc69_distance', 'k_jc69_mfma': 'jc69_distance', 'k_jc69': 'jc69_distance', 'k_scoredist': 'scoredist_distance', 'k_select_fast': 'select_fast', 'k_select': 'select',
         'k_sweep': 'lsq_sweep'}
res = {}
for k, v in summ.items():
    if 'FETCH_SIZE' not in v or v.get('mean_ns_under_pmc', 0) < 20000:
        continue
    key = next((names[n] for n in sorted(names, key=len, reverse=True) if n in k), None)
    if key is None:
        continue
    if key == 'lsq_sweep' and '64>' not in k and v['mean_ns_under_pmc'] < 1e5:
        continue
    fetch, write = v['FETCH_SIZE'] * 1024, v['WRITE_SIZE'] * 1024
    streaming = key != 'lsq_sweep'
    e = {'kernel': k, 'fetch_bytes_raw': fetch, 'write_bytes': write,
         'hbm_bytes_per_launch': (2 * fetch if streaming else fetch) + write,
         'fetch_correction': 'x2 (wide coalesced streaming reads)' if streaming else 'raw (scattered 64-B records: uncalibrated; x2 would give %d)' % (2 * fetch + write),
         'mean_ns_under_pmc': v['mean_ns_under_pmc'], 'l2_hit_rate': v['TCC_HIT_sum'] / max(v['TCC_HIT_sum'] + v['TCC_MISS_sum'], 1),
         'tcc_read_req': v.get('TCC_READ_sum'), 'tcc_write_req': v.get('TCC_WRITE_sum'), 'tcc_atomic_req': v.get('TCC_ATOMIC_sum')}
    if key not in res or e['mean_ns_under_pmc'] > res[key]['mean_ns_under_pmc']:
        res[key] = e
allj = {}
if os.path.exists(out):
    allj = json.load(open(out))
allj[workload] = res
json.dump(allj, open(out, 'w'), indent=1, sort_keys=True)
print(json.dumps(res, indent=1))
