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
names = {'k_jc69_gemm': 'jc69_distance', 'k_jc69_mfma': 'jc69_distance', 'k_jc69': 'jc69_distance', 'k_scoredist': 'scoredist_distance', 'k_sd_gemm': 'scoredist_filter_gemm', 'k_sd_exact': 'scoredist_exact', 'k_sd_topup': 'scoredist_topup',
         'k_select_fast': 'select_fast',
         'k_select_stream': 'table_select' if workload == 'c5' else 'select', 'k_select_clusters': 'select_clusters', 'k_cluster_dist': 'cluster_dist', 'k_select': 'select',
         'k_blocks_up': 'blocks_up', 'k_blocks_down': 'blocks_down',
         'k_sweep_mixed': 'lsq_sweep', 'k_sweep<': 'lsq_sweep', 'k_lean_up': 'lsq_sweep_up', 'k_lean_down': 'lsq_sweep_down',
         'k_sweep_lean_big': 'lsq_sweep_big'}
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
    streaming = not key.startswith('lsq_sweep')  # (the block kernels read and write whole 512-byte rows: streaming)
    e = {'kernel': k, 'fetch_bytes_raw': fetch, 'write_bytes': write,
         'hbm_bytes_per_launch': (2 * fetch if streaming else fetch) + write,
         'fetch_correction': 'x2 (wide coalesced streaming reads)' if streaming else 'raw (scattered / short runs: uncalibrated; x2 would give %d)' % (2 * fetch + write),
         'mean_ns_under_pmc': v['mean_ns_under_pmc'], 'dispatches_per_pass': v.get('dispatches_per_pass'),
         'l2_hit_rate': v['TCC_HIT_sum'] / max(v['TCC_HIT_sum'] + v['TCC_MISS_sum'], 1),
         'tcc_read_req': v.get('TCC_READ_sum'), 'tcc_write_req': v.get('TCC_WRITE_sum'), 'tcc_atomic_req': v.get('TCC_ATOMIC_sum')}
    if key not in res or e['mean_ns_under_pmc'] > res[key]['mean_ns_under_pmc']:
        res[key] = e
# sweep_lean.hip runs a device batch's sweep as three kernels (bottom-up, top-down, workgroup-sized teams: the last one's
# figures are the mean over its dispatches, half of which are empty overflow launches -> doubled): one figure for the batch
parts = [res[k] for k in ('lsq_sweep_up', 'lsq_sweep_down', 'blocks_down') if k in res]  # (clustered route: the sweep phase's timer holds k_blocks_down too)
if parts and 'lsq_sweep' not in res:
    big = res.get('lsq_sweep_big')
    res['lsq_sweep'] = {'kernel': 'k_lean_up + k_lean_down + k_sweep_lean_big' + (' + k_blocks_down' if 'blocks_down' in res else '') + ' (per device batch)',
                        'hbm_bytes_per_launch': sum(p_['hbm_bytes_per_launch'] for p_ in parts) + (2 * big['hbm_bytes_per_launch'] if big else 0),
                        'mean_ns_under_pmc': sum(p_['mean_ns_under_pmc'] for p_ in parts) + (2 * big['mean_ns_under_pmc'] if big else 0),
                        'fetch_correction': 'raw', 'note': 'serialised under the counter passes; the three run side by side in the bench'}
allj = {}
if os.path.exists(out):
    allj = json.load(open(out))
allj[workload] = res
json.dump(allj, open(out, 'w'), indent=1, sort_keys=True)
print(json.dumps(res, indent=1))
