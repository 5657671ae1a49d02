#!/usr/bin/env python3
"""Benchmark of the APPLES per-query hot path on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5|c3-clustered|c4-clustered|small]
                    [--scaling weak|strong] [--no-cpu] [--queries Q]

Default workload = BASELINE.json config 3, the one the targets are quoted on: synthetic 200 k-leaf
backbone, L = 1000 nt, 100 k queries, OLS/JC69, -f 0.2 -b 25.

A step = one pass of the hot path over one block of synthetic queries, timed the way the reference
times its own "Processed all queries" interval (run_apples.py:93-104): the queries start as byte
arrays in host memory (the reference's S1 arrays) and the step ends when the placements are back in
host memory -- upload, packing into the device layouts, distance vector, observed-set selection,
least-squares sweep, best edge, copy back (apples_place_from_sequences).  The packed reference
alignment and the tree are resident in HBM (the reference's workers likewise hold them before the
timer starts).  value = query placements/s for the whole job.  `resident` in the JSON line is the
same pass with the query block already uploaded and packed (device time only).

For N > 1 there is one rank process per GPU: `python bench.py --gpus N` starts them itself (a parent that never
touches a device spawns N fresh interpreters with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and relays rank
0's line: apples_amd/launcher.py, in place of run_apples.py:93-102's fork pool), and under torch.distributed.run
(RANK already in the environment) the process is a rank.  Every rank places its own shard with no collective in
the data path and the 40-byte placement structs are gathered to rank 0 with one RCCL gather inside the timed
region (--gather rccl, the default: ctypes on librccl.so, no PyTorch in the process; --gather torch:
torch.distributed, backend nccl).  --scaling strong (default for c3, BASELINE config 3's "100 k queries, 1 -> 8
GPUs"): the workload's queries are split over the ranks; --scaling weak (default elsewhere): every rank places a
block of the workload's size.  Fewer visible devices than ranks is an error, not an oversubscription.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, algorithmic bytes or operations /
HIP-event time), `roofline_hbm_point` (BASELINE.json's second figure: the distance kernel at one
query per pass over a packed reference larger than every cache, streamed from HBM for several seconds after the
timed region), `distance_kernel_stream` (the same kernel on the workload's own reference, which the Infinity
Cache holds: the cache figure), `strong_scaling_proxy` (the 8-GPU shards of the query set timed one by one on
this GPU), `clustered` (the command line's default route on the same inputs) and
`cpu_baseline` (the CPU restatement of the reference path -- per-query numpy/Python worker under a
fork pool, as run_apples.py:101-102 -- timed on this host's cores on a bounded sample of the same
workload).
"""
import os

# the CPU baseline's workers are single-threaded processes, as the reference's are
os.environ.setdefault('OMP_NUM_THREADS', '1')
os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
os.environ.setdefault('MKL_NUM_THREADS', '1')

import argparse  # noqa: E402
import zlib  # noqa: E402
import json  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (leaves, L, queries, protein, method, threshold)   -- BASELINE.json configs
    'c2': (10000, 1000, 10000, False, 'OLS', 0.2),
    'c3': (200000, 1000, 100000, False, 'OLS', 0.2),
    'c4': (50000, 500, 50000, True, 'FM', 0.2),
    'small': (2000, 500, 2048, False, 'OLS', 0.2),
    # C5: distance-table input (-d), least-squares-only path; L is irrelevant (no alignment).  100 k rows of
    # 200 k fp64 columns are 160 GB: the bench holds a block of 4 096 rows (6.5 GB) unless --queries says otherwise
    'c5': (200000, 0, 4096, False, 'BME', 0.2),
    # the command line's default route at C3 size: max-diameter clusters at 1.2 x -f with consensus
    # representatives (apples/Reference.py:84-157) instead of all-singleton clusters
    'c3-clustered': (200000, 1000, 100000, False, 'OLS', 0.2),
    # ... and at C4's: the default route of `-p` (scoredist to consensus representatives of the 21-symbol alphabet)
    'c4-clustered': (50000, 500, 50000, True, 'FM', 0.2),
    # SURVEY 8d's stress variant: -f 1e9, every leaf observed (V = 2 N - 2 nodes swept per query: the worst case for both
    # kernel families); C3's shape on a bounded sample of its queries
    'c2-all': (10000, 1000, 10000, False, 'OLS', 1e9),
    'c3-all': (200000, 1000, 2048, False, 'OLS', 1e9),
}
MFMA_F4_PEAK_TOPS = 10000.0  # dense fp4 peak at the nominal clock, /opt/skills/guides/MI355X_MICROARCH.md, matrix cores table
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
LDS_PEAK_GBS = 150000.0  # ds_read_b64/b128 with every CU streaming (MI355X_MICROARCH.md, LDS section)


def _spin(deadline):
    t0 = time.process_time()
    x = 0
    while time.time() < deadline:
        x += 1
    return time.process_time() - t0


def _cores():
    """(processes the CPU leg should use, logical CPUs in the affinity mask, distinct physical cores among them).
    The affinity mask is not what a container may use: a CPU quota (cgroup cpu.max) throttles a pool of
    one process per logical CPU to a small fraction of them.  So the usable parallelism is measured: one
    spinning process per CPU the quota allows, CPU seconds obtained / wall seconds."""
    import multiprocessing as mp
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else list(range(os.cpu_count() or 1))
    phys = set()
    for c in cpus:
        try:
            with open('/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list' % c) as f:
                phys.add(f.read().strip())
        except OSError:
            phys.add(str(c))
    n = len(cpus)
    quota = n
    for path, parse in (('/sys/fs/cgroup/cpu.max', lambda t: t.split()),
                        ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', lambda t: [t.strip(), open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read()])):
        try:
            q, per = parse(open(path).read())
            if q != 'max' and float(q) > 0:
                quota = min(quota, max(1, int(float(q) / float(per) + 0.5)))
        except (OSError, ValueError):
            pass
    n = quota
    eff = 1
    with mp.get_context('fork').Pool(n) as pool:
        pool.map(_spin, [time.time() + 0.3] * n)       # start-up out of the way
        for _ in range(3):                             # best of three windows (a VM may need a moment to spread the pool)
            t0 = time.time()
            got = sum(pool.map(_spin, [t0 + 0.5] * n, chunksize=1))
            eff = max(eff, int(round(got / max(time.time() - t0, 1e-3))))
    return max(1, min(n, eff)), len(cpus), len(phys)


def cpu_baseline(ds, protein, method, threshold, target_cpu_seconds=10.0):
    """Time the oracle's pool driver on a bounded, seeded sample of the same queries."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import apples_oracle as orc
    cores, logical, phys = _cores()
    kw = dict(protein=protein, method=method, criterion='MLSE', threshold=threshold, baseobs=25, overlap=0.001)
    # one core alone: three queries in process
    dt1, _, _ = orc.time_pool(ds.tree, ds.ref_names, ds.ref_seqs, ds.query_names[:3], ds.query_seqs[:3], 1, **kw)
    t1 = max(dt1 / 3.0, 1e-4)
    # every process gets the same number of queries: about target_cpu_seconds of wall time if the pool scaled
    # perfectly.  The pool is started and warmed before the clock starts (start-up, which the reference's
    # own "Processed all queries" timer would include, is reported apart)
    per_core = int(max(1, min(64, target_cpu_seconds / t1)))
    n = int(min(len(ds.query_names), per_core * cores))
    dt, startup, _ = orc.time_pool(ds.tree, ds.ref_names, ds.ref_seqs, ds.query_names[:n], ds.query_seqs[:n], cores, **kw)
    ideal = cores / t1
    return {'value': n / dt, 'unit': 'queries/s', 'cores': cores, 'kind': 'port',
            'logical_cpus': logical, 'physical_cores': phys, 'single_core_queries_per_s': 1.0 / t1,
            'single_core_x_cores': ideal, 'pool_efficiency': (n / dt) / ideal,
            'sample': 'first %d of the %d synthetic queries on a warmed %d-process fork pool, single-threaded numpy (the '
                      'host shows %d logical CPUs / %d physical cores; %d is the parallelism a spinning pool actually '
                      'obtains here): %.2f s steady state (pool start-up %.1f s not counted); one process alone places '
                      '%.3f queries/s, x %d = %.1f: the pool reaches %.0f %% of that'
                      % (n, len(ds.query_names), cores, logical, phys, cores, dt, startup, 1.0 / t1, cores, ideal,
                         100.0 * (n / dt) / ideal)}


def cpu_baseline_table(ds, D, method, threshold, target_cpu_seconds=10.0):
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import apples_oracle as orc
    cores, logical, phys = _cores()
    t0 = time.time()
    orc.time_pool_table(ds.tree, ds.ref_names, ds.query_names[:1], D[:1], 1, method=method, threshold=threshold)
    t1 = max(time.time() - t0, 1e-3)
    n = int(min(len(D), max(cores, min(64 * cores, cores * target_cpu_seconds / t1))))
    dt, startup, _ = orc.time_pool_table(ds.tree, ds.ref_names, ds.query_names[:n], D[:n], cores, method=method,
                                         threshold=threshold)
    return {'value': n / dt, 'unit': 'queries/s', 'cores': cores, 'kind': 'port', 'logical_cpus': logical, 'physical_cores': phys,
            'sample': 'first %d table rows on a warmed %d-process fork pool (%d logical CPUs; %d = measured usable '
                      'parallelism): %.2f s steady state (start-up %.1f s not counted; %.3f s for one query on one core '
                      'incl. pool start)' % (n, cores, logical, cores, dt, startup, t1)}


def distance_stream_point(eng, ds, L):
    """BASELINE.json's second figure, measured live: the distance kernel at query tile T = 1 (SURVEY
    8d's roofline point: the packed reference streamed once per query, full fp64 rows out), on up to
    256 of the workload's queries, outside the timed region.  `delivered` counts the bytes the kernel
    moves (packed reference + 8 B per pair); whether they come from HBM or from the caches depends
    on the reference's size, which is reported beside it (profiles/ holds the HBM-only point on a
    reference larger than every cache, with its FETCH_SIZE counter pass)."""
    n = min(256, len(ds.query_seqs))
    h, n = eng.upload_queries(ds.query_seqs[:n])
    for _ in range(3):
        eng.distances_resident(h, 1)
    ms = eng.timing()['dist_ms']
    info = eng.describe()
    eng.free_queries(h)
    if ms <= 0:
        return None
    moved = n * (info['packed_bytes'] + info['n_rows'] * 8.0)
    algo = n * (info['n_rows'] * (L + 8.0) + L)
    gbs = moved / (ms * 1e-3) / 1e9
    return {'query_tile': 1, 'queries': n, 'ms': ms, 'packed_reference_bytes': info['packed_bytes'],
            'delivered_GBps': gbs, 'peak_GBps': HBM_PEAK_GBS, 'frac': gbs / HBM_PEAK_GBS,
            'algorithmic_GBps': algo / (ms * 1e-3) / 1e9,
            'served_from': 'L2 / Infinity Cache (reference smaller than the 256 MiB cache): not an HBM measurement'
                           if info['packed_bytes'] < (256 << 20) else 'HBM'}


def hbm_point(ref_seqs, queries, device, seconds=5.0, min_packed_bytes=1100 << 20):
    """BASELINE.json's second figure where no cache can serve it: the distance kernel at query tile T = 1 (SURVEY
    8d's roofline point: the packed reference streamed once per query, full fp64 rows out) over a reference made of
    the workload's rows repeated until its packed form exceeds 1 GiB (four times the 256 MiB Infinity Cache), launched
    back to back for `seconds` of continuous kernel time.  delivered = (packed reference + 8 B per pair) per query /
    HIP-event time of the launch (median over the launches)."""
    from apples_amd.engine import Engine
    from apples_amd.tree import parse_newick
    n0, L = ref_seqs.shape
    bytes_per_row = 12 * ((L + 31) // 32)           # gap plane + two code planes per 32-site word
    reps = max(1, -(-min_packed_bytes // (bytes_per_row * n0)))
    ref = np.ascontiguousarray(np.tile(ref_seqs, (reps, 1)))
    nq = min(256, len(queries))
    tree = parse_newick('((A:0.1,B:0.2):0.25,(C:0.3,(D:0.2,E:0.2):0.2):0.25);')
    eng = Engine(tree, ref, np.full(len(ref), -1, np.int32), method='OLS', max_batch=nq, device=device)
    del ref
    try:
        h, n = eng.upload_queries(queries[:nq])
        eng.distances_resident(h, 1)
        ms = []
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            eng.distances_resident(h, 1)
            ms.append(eng.timing()['dist_ms'])
        wall = time.perf_counter() - t0
        info = eng.describe()
        eng.free_queries(h)
    finally:
        eng.close()
    med = float(np.median(ms))
    moved = n * (info['packed_bytes'] + info['n_rows'] * 8.0)
    gbs = moved / (med * 1e-3) / 1e9
    return {'kernel': 'k_jc69 (bit planes), query tile 1, full fp64 rows out', 'what': 'DELIVERED bandwidth (bytes the kernel moves / time): '
            'FETCH_SIZE counts Infinity-Cache hits and the 256 queries of a launch march over the same reference together, so this is not '
            'a pure HBM figure (the guide\'s measured-achievable HBM rate is ~6.3 TB/s)',
            'rows': int(info['n_rows']), 'L': int(L),
            'queries_per_launch': int(n), 'launches': len(ms), 'continuous_kernel_seconds': float(sum(ms) * 1e-3),
            'wall_seconds': wall, 'ms_per_launch_median': med, 'us_per_query': med * 1e3 / n,
            'packed_reference_bytes': int(info['packed_bytes']), 'bytes_moved_per_launch': moved,
            'delivered_GBps': gbs, 'peak_GBps': HBM_PEAK_GBS, 'frac_of_8000': gbs / HBM_PEAK_GBS,
            'frac_of_6300_achievable': gbs / 6300.0,
            'algorithmic_GBps': n * (info['n_rows'] * (L + 8.0) + L) / (med * 1e-3) / 1e9,
            'served_from': 'HBM: the packed reference (%d MiB) is larger than the 256 MiB Infinity Cache and is streamed whole '
                           'for every query' % (info['packed_bytes'] >> 20)}


def strong_scaling_proxy(eng, queries, ms_full, parts=8, reps=3, device=0, full=None):
    """What a single GPU can say about BASELINE config 3's 1 -> 8 GPU curve: the `parts` contiguous shards of the one
    query set (apples_amd/distributed.py:shard_bounds, what rank r of an 8-GPU job places) timed one after the other,
    host buffer -> placements in host memory like the step itself.  The 8-GPU step ends when its slowest rank does
    (observed sets are heavy-tailed), so the prediction uses the slowest shard; the gather of
    40 B per query over xGMI is not in it."""
    from apples_amd.distributed import shard_bounds
    ms = []
    differing = []
    for lo, hi in shard_bounds(len(queries), parts):
        block = np.ascontiguousarray(queries[lo:hi])
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            got = eng.place_sequences(block)
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        # (a shard is a small device batch: lower routing cut, 512-thread routed teams, the top-up chain beside the sweep --
        # other routes than the full set's batches took, the same bytes)
        if full is not None and got.tobytes() != full[lo:hi].tobytes():
            differing.append([int(lo), int(hi), int((got['edge'] != full[lo:hi]['edge']).sum())])  # (reported in the line, never hidden)
        ms.append(best)
    # the end-of-run gather as far as one GPU can measure it: the torch-free RCCL path (apples_amd/rccl.py, world size 1: a
    # grouped ncclSend / ncclRecv to self + the copy to the host) on the whole job's 40-byte structs -- what rank 0 of an
    # 8-GPU job receives and brings to the host; the other ranks' sends travel over seven xGMI links at once
    gather_ms = None
    h = comm = None
    try:
        from apples_amd.rccl import Comm
        h, n = eng.place_sequences_streamed(queries)
        comm = Comm(0, 1, device)
        g = []
        for _ in range(5):
            t0 = time.perf_counter()
            comm.gather_to_host(eng.placements_device_ptr(h), [n * 40])
            g.append((time.perf_counter() - t0) * 1e3)
        gather_ms = min(g)
    except Exception as e:  # no librccl on this host: say so in the line
        gather_note = 'gather not measured here: %s' % e
    finally:  # whatever happened: the resident block and the communicator do not stay behind to skew the legs that follow
        if comm is not None:
            comm.close()
        if h is not None:
            eng.free_queries(h)
    worst = max(ms) + (gather_ms or 0.0)
    return {'parts': parts, 'queries_per_shard': [b - a for a, b in shard_bounds(len(queries), parts)],
            'ms_shard': ms, 'ms_shard_max': max(ms), 'ms_shard_mean': float(np.mean(ms)), 'ms_full_set': ms_full,
            'gather_ms': gather_ms, 'shards_equal_full_set': (not differing) if full is not None else None,
            'shards_that_differ': differing,  # [first query, last + 1, placements with another edge] per shard whose bytes differ

            'predicted_speedup_at_%d' % parts: ms_full / worst,
            'note': 'A PREDICTION, not a measurement (this pool has one-GPU boxes): shards of the one query set timed one by one '
                    'on this GPU (best of %d, host buffer -> host); predicted speed-up = full-set step / (slowest shard + gather_ms); '
                    'gather_ms = %s' % (reps, 'world-1 RCCL self-gather of all %d placement structs + copy to the host, best of 5'
                                        % len(queries) if gather_ms is not None else gather_note)}


def clustered_leg(ds, nodes, queries, thr, method, device, steps=3):
    """The command line's default route on the same inputs (run_apples.py without --no-clusters, as the reference's
    default, apples/Reference.py:84-157): max-diameter clusters at 1.2 x -f with consensus representatives."""
    from apples_amd.engine import Engine
    t0 = time.perf_counter()
    clusters = make_clusters(ds, thr)
    t_clusters = time.perf_counter() - t0
    eng = Engine(ds.tree, ds.ref_seqs, nodes, clusters=clusters, protein=False, method=method, criterion='MLSE',
                 threshold=thr, baseobs=25, overlap=0.001, device=device)
    try:
        out = eng.place_sequences(queries)
        ph = {'dist_ms': 0.0, 'select_ms': 0.0, 'sweep_ms': 0.0, 'blocks_ms': 0.0}
        t0 = time.perf_counter()
        for _ in range(steps):
            out = eng.place_sequences(queries)
            t = eng.timing()
            for k in ph:
                ph[k] += t[k]
        dt = (time.perf_counter() - t0) / steps
        info = eng.describe()
    finally:
        eng.close()
    return {'value': len(queries) / dt, 'unit': 'queries/s', 'ms_per_step': dt * 1e3, 'steps': steps,
            'per_kernel_ms_per_step': {k: v / steps for k, v in ph.items()}, 'n_reps': int(info['n_reps']),
            'mean_observed': float(np.mean(out['n_obs'])), 'placed': int((out['n_valid'] > 0).sum()),
            'cluster_blocks': int(info.get('cluster_blocks', 0)), 'device_batch': int(info['batch']),
            'clustering_and_consensus_s': t_clusters,
            'note': 'same tree, reference and queries through max-diameter clusters at 1.2 x -f with consensus representatives '
                    '(the route run_apples.py takes by default); host buffer -> placements in host memory.  blocks_ms = k_blocks_up (the '
                    'sweep inside the clade blocks, bottom-up), which runs beside the last phase of select_ms; k_blocks_down is in sweep_ms'}


def roofline_of(workload, nq, rows, L, protein, table, per_step, launches_per_step, placements, info, filter_ms=None):
    """`roofline` of the dominant kernel (what bounds each kernel and its algorithmic bytes / operations
    per unit): algorithmic work per launch / HIP-event time per launch."""
    placed = placements['n_valid'] > 0
    # algorithmic bytes per step: distance N_rows*(L+8)+L per query; sweep 332*V per query
    dist_bytes = nq * (rows * (L + 8) + L)
    if info.get('cluster_fused'):
        # the clustered route's pairs are (query, representative) and (query, member of an accepted cluster) -- what
        # apples/Reference.py:138-152 computes: pricing it by N_rows pairs per query would report work nobody asks for
        dist_bytes = nq * ((float(info.get('n_reps') or 0) + float(np.mean(placements['n_obs']))) * (L + 8) + L)
    sweep_bytes = 332.0 * float(np.sum(placements['n_valid'][placed] + 1))
    # (clustered route with clade blocks: the sweep inside the blocks starts in k_blocks_up, which runs beside the selection's last
    # phase -- its own timer, blocks_ms; k_blocks_down is inside sweep_ms)
    kernels = {'lsq_sweep': (sweep_bytes, per_step['sweep_ms'] + per_step.get('blocks_ms', 0.0))}
    if table:  # the -d filter reads every table value once
        kernels['table_select'] = (nq * rows * 8.0, per_step['select_ms'])
    else:
        kernels['jc69_distance' if not protein else 'scoredist_distance'] = (dist_bytes, per_step['dist_ms'])
    # (the overlapped part of a phase is not its own: k_blocks_up runs beside the selection's last phase and its time is inside
    # select_ms too -- it never makes the sweep the dominant kernel by itself)
    dom = max(kernels, key=lambda k: kernels[k][1] - (per_step.get('blocks_ms', 0.0) if k == 'lsq_sweep' else 0.0))
    achieved = kernels[dom][0] / (kernels[dom][1] * 1e-3) / 1e9 if kernels[dom][1] > 0 else 0.0
    traffic, traffic_commit = load_traffic(workload, dom)
    # what the numbers of this line can be checked against in profiles/ (rocprofv3 --kernel-trace --stats of this command): the step
    # is cut into `device_batches` batches of `batch_queries` queries (apples_describe: the workspace's batch; equal batches); the
    # dominant kernel is launched `kernel_calls_per_step` times per step (the first batch of a host buffer in two pieces) and takes
    # `dominant_kernel_ms_per_step` in all: the kernel's total time in the trace / the passes traced
    batch = int(info.get('batch') or 0)
    n_batches = max(1, -(-nq // batch)) if batch > 0 else 1
    roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_measured_at_commit': traffic_commit,
                'traffic_is': 'HBM bytes per launch of the dominant kernel, mean over the launches of the committed counter passes '
                              '(profiles/pmc_summary.json: separate --pmc runs of this command, FETCH_SIZE corrected as the guide prescribes)',
                'device_batches': n_batches, 'batch_queries': -(-nq // n_batches),
                'kernel_calls_per_step': launches_per_step, 'dominant_kernel_ms_per_step': kernels[dom][1],
                'algorithmic_bytes_per_step': kernels[dom][0],
                'per_kernel_ms_per_step': per_step,
                'all_kernels_GBps': {k: (v[0] / (v[1] * 1e-3) / 1e9 if v[1] > 0 else 0.0) for k, v in kernels.items()}}
    # the sweep against HBM both ways: SURVEY 8d's algorithmic 332 B per swept node, and what the counters saw it move
    sw_bytes, _ = load_traffic(workload, 'lsq_sweep')
    sweep_t = kernels['lsq_sweep'][1]
    if sweep_t > 0:
        roofline['sweep_hbm'] = {'algorithmic_GBps': kernels['lsq_sweep'][0] / (sweep_t * 1e-3) / 1e9,
                                 'algorithmic_frac': kernels['lsq_sweep'][0] / (sweep_t * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 'counter_bytes_per_device_batch': sw_bytes,
                                 'counter_GBps': (sw_bytes * n_batches / (sweep_t * 1e-3) / 1e9) if sw_bytes else None,
                                 'counter_frac': (sw_bytes * n_batches / (sweep_t * 1e-3) / 1e9 / HBM_PEAK_GBS) if sw_bytes else None,
                                 'note': 'counter figure = HBM bytes of one device batch\'s sweep kernels (committed PMC passes) x device_batches / sweep_ms'}
    if dom == 'lsq_sweep' and per_step.get('blocks_ms', 0.0) > 0:
        roofline['note'] = ('clustered route with clade blocks: time = sweep_ms + blocks_ms (k_blocks_up runs beside the selection); the '
                            'algorithmic figure is SURVEY 8d\'s 332 B per swept node, the block kernels move ~190 B per block-internal node '
                            '(tuples of [slot][component][lane] rows shared structure): a fraction above 1 means fewer bytes moved, see traffic')
    if dom == 'jc69_distance' and info.get('code_planes') == 2 and info.get('all_singleton') and \
            not os.environ.get('APPLES_NO_DIST_MFMA') and not os.environ.get('APPLES_NO_FUSE'):
        # the tiled pair counts run on the matrix cores (fp4 operands, 4 MACs per site and pair:
        # DESIGN.md section 4): price them against the dense fp4 MFMA peak, 2 ops per MAC
        ops = 2.0 * 4.0 * nq * rows * 32.0 * ((L + 31) // 32)
        tops = ops / (kernels[dom][1] * 1e-3) / 1e12
        roofline.update({'bound': 'mfma', 'achieved': tops, 'peak': MFMA_F4_PEAK_TOPS, 'unit': 'TFLOP/s',
                         'frac': tops / MFMA_F4_PEAK_TOPS, 'algorithmic_ops_per_step': ops,
                         'executed_vs_survey_ops': 'executed fp4 operations = 2 x 4 components x 32-site words: 4.1 x SURVEY 8d\'s 2 L byte-operations per pair '
                                                   '(three tetrahedron components + validity: the smallest exact bilinear form of the two counts)',
                         'hbm_algorithmic_GBps': achieved,
                         'launch_note': 'the first device batch of a host buffer is two launches (its first quarter, while the rest is still on the '
                                        'bus, then the rest); the resident pass launches whole batches'})
    elif dom == 'scoredist_distance' and info.get('scoredist_filter') and filter_ms:
        # the fused scoredist pass = a lower bound of every pair's table sum on the matrix cores (fp4 operands, 20 values per
        # site: dist_sd.hip) + the exact evaluation of the ~1 % of the pairs that survive it; the dominant kernel is the
        # matrix-core filter: 2 x 20 L operations per pair in 128-value steps, priced against the dense fp4 peak
        steps = (20 * L + 127) // 128
        pad = (rows + 255) // 256 * 256
        ops = 2.0 * nq * pad * steps * 128.0
        tops = ops / (filter_ms * 1e-3) / 1e12
        roofline.update({'bound': 'mfma', 'kernel': 'scoredist_filter_gemm', 'achieved': tops, 'peak': MFMA_F4_PEAK_TOPS, 'unit': 'TFLOP/s',
                         'frac': tops / MFMA_F4_PEAK_TOPS, 'algorithmic_ops_per_step': ops,
                         'dominant_kernel_ms_per_step': filter_ms, 'hbm_algorithmic_GBps': achieved,
                         'note': 'dist_ms = filter (filter_ms) + exact evaluation of its candidates; the table look-ups the '
                                 'reference makes for every pair (8 B x 20 x L x pairs from LDS: the round-3 bound) are made for the candidates only'})
        roofline['traffic'], roofline['traffic_measured_at_commit'] = load_traffic(workload, 'scoredist_filter_gemm')
    elif dom == 'scoredist_distance':
        # one 8-byte table read from LDS per site and pair: the LDS read rate bounds it
        pairs = (float(info.get('n_reps') or 0) + float(np.mean(placements['n_obs']))) if info.get('cluster_fused') else rows  # (per query: what the route computes)
        lds = 8.0 * nq * pairs * L / (kernels[dom][1] * 1e-3) / 1e9
        roofline.update({'bound': 'lds', 'achieved': lds, 'peak': LDS_PEAK_GBS, 'frac': lds / LDS_PEAK_GBS,
                         'hbm_algorithmic_GBps': achieved})
    return roofline


def variant_dataset(ds, variant, n_leaves, L, Q, protein):
    """Config 3's inputs in the shapes real inputs have beside the strictly binary, shallow, ACGT--only synthetic set:
    'unrooted' (the same tree with its root trifurcated: what FastTree prints and what both of the reference's example
    backbones look like), 'polytomies' (1 % of the internal nodes dissolved into their parents), 'deep' (the random-join
    subtrees hung on a caterpillar spine of 400 nodes: more than 254 levels; its own alignment), 'dots' (one `.` per 1 000
    sites in 5 % of the reference rows and one query with a `?`: bytes beyond ACGT-, ordinary symbols to
    apples/distance.py:733-737), 'L4000' (every row four times over: the same distances from four times the sites),
    '300k' (300 000 leaves: its own alignment)."""
    import copy
    from apples_amd import synth
    if variant == 'deep':
        return synth.make_dataset(n_leaves, L, Q, protein=protein, spine=400)
    if variant == '300k':
        return synth.make_dataset(300000, L, Q, protein=protein)
    d = copy.copy(ds if ds is not None else synth.make_dataset(n_leaves, L, Q, protein=protein))
    if variant in ('unrooted', 'polytomies'):
        d.newick = synth.reshape_newick(d.tree, variant)
        d.tree = synth.parse_newick(d.newick)
    elif variant == 'dots':
        rng = np.random.default_rng(17)
        d.ref_seqs = d.ref_seqs.copy()
        rows = np.nonzero(rng.random(len(d.ref_seqs)) < 0.05)[0]
        for k in range((L + 999) // 1000):
            lo, hi = 1000 * k, min(L, 1000 * (k + 1))
            d.ref_seqs[rows, rng.integers(lo, hi, size=len(rows))] = ord('.')
        d.query_seqs = d.query_seqs.copy()
        d.query_seqs[0, int(rng.integers(0, L))] = ord('?')
    elif variant == 'L4000':
        d.ref_seqs = np.ascontiguousarray(np.tile(d.ref_seqs, (1, 4)))
        d.query_seqs = np.ascontiguousarray(np.tile(d.query_seqs[:Q], (1, 4)))
    elif variant:
        raise ValueError(variant)
    return d


def other_workload(name, device, steps=3, ds=None, queries=0, variant=None, prepared=False):
    """A compact leg of another BASELINE config for the driver's line: `steps` timed passes (host buffers -> placements in
    host memory; config 5: the table block resident), per-kernel HIP-event times, the dominant kernel's roofline.  No CPU leg.
    variant: see variant_dataset (config 3's size in the shapes the fast routes used to refuse)."""
    from apples_amd import synth
    from apples_amd.engine import Engine
    n_leaves, L, Q, protein, method, thr = WORKLOADS[name]
    if queries:
        Q = queries
    table = name == 'c5'
    clustered = name.endswith('-clustered')  # the command line's default route: clusters + consensus representatives
    if variant:  # (prepared: ds already is the variant's dataset)
        ds_ = ds if prepared else variant_dataset(ds, variant, n_leaves, L, Q, protein)
        n_leaves, L = len(ds_.ref_names), ds_.ref_seqs.shape[1]
    elif ds is None or table:
        ds_ = synth.make_dataset(n_leaves, L if not table else 4, Q, protein=protein) if ds is None else ds
    else:
        ds_ = ds
    nodes = np.array([ds_.tree.name_to_node[n] for n in ds_.ref_names], np.int32)
    if table:
        index = synth.TreeIndex(ds_.tree)
        rs = np.random.default_rng(3)
        q_leaf = rs.integers(0, n_leaves, size=Q)
        q_pend = rs.exponential(0.01, size=Q)
        D = np.empty((Q, n_leaves))
        for lo in range(0, Q, 2048):  # (block by block: the generator's temporaries stay small)
            hi = min(Q, lo + 2048)
            D[lo:hi] = synth.fast_distance_rows(ds_.tree, index, q_leaf, q_pend, list(range(lo, hi)), seed_noise=7 + lo)
        eng = Engine(ds_.tree, None, method=method, criterion='MLSE', threshold=thr, baseobs=25, device=device)
        handle, _ = eng.upload_table(D, nodes)

        def step():
            eng.place_resident(handle)
            return eng.fetch(handle, Q)
    else:
        eng = Engine(ds_.tree, ds_.ref_seqs, nodes, clusters=make_clusters(ds_, thr, protein) if clustered else None, protein=protein,
                     method=method, criterion='MLSE', threshold=thr, baseobs=25, overlap=0.001, device=device)
        qarr = np.ascontiguousarray(ds_.query_seqs[:Q])

        def step():
            return eng.place_sequences(qarr)
    try:
        out = step()
        ph = {'dist_ms': 0.0, 'select_ms': 0.0, 'sweep_ms': 0.0, 'filter_ms': 0.0, 'blocks_ms': 0.0}
        launches = 0
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
            tm = eng.timing()
            for k in ph:
                ph[k] += tm[k]
            launches += tm['dist_launches']
        dt = (time.perf_counter() - t0) / steps
        info = eng.describe()
    finally:
        eng.close()
    per = {k: ph[k] / steps for k in ('dist_ms', 'select_ms', 'sweep_ms')}
    if ph['blocks_ms'] > 0:
        per['blocks_ms'] = ph['blocks_ms'] / steps
    rf = roofline_of(name, Q, n_leaves if table else info['n_rows'], L, protein, table, per, launches / steps, out, info,
                     filter_ms=ph['filter_ms'] / steps)
    return {'value': Q / dt, 'unit': 'queries/s', 'ms_per_step': dt * 1e3, 'steps': steps, 'queries': Q,
            'timed': 'table block resident -> placements in host memory' if table else 'host byte arrays -> placements in host memory',
            'per_kernel_ms_per_step': per,
            # (the committed counter passes are of the workload's own size: no traffic figure for another number of queries)
            'roofline': {k: (None if k == 'traffic' and queries else rf[k])
                         for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'dominant_kernel_ms_per_step', 'kernel_calls_per_step',
                                   'device_batches', 'batch_queries', 'traffic') if k in rf},
            'mean_observed': float(np.mean(out['n_obs'])), 'placed': int((out['n_valid'] > 0).sum()),
            **({'n_reps': int(info['n_reps']), 'cluster_fused': int(info.get('cluster_fused', 0)),
                'cluster_blocks': int(info.get('cluster_blocks', 0))} if clustered else {}),
            # which route served it (apples_describe): the legs of the other input shapes are there to show this
            **({'variant': variant,
                'route': {k: info.get(k) for k in ('sweep_layout', 'fused_distance_pass', 'code_planes', 'cluster_fused', 'cluster_blocks',
                                                   'height', 'max_children', 'n_rows', 'length', 'batch')}} if variant else {})}


def load_traffic(workload, kernel):
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 PMC passes
    (profiles/pmc_summary.json, written by scripts/pmc_to_traffic.py from separate --pmc runs of
    this same command; FETCH_SIZE/WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes), with the
    commit the counters were collected at."""
    path = os.path.join(ROOT, 'profiles', 'pmc_summary.json')
    try:
        with open(path) as f:
            d = json.load(f)
        e = d[workload][kernel]
        return e['hbm_bytes_per_launch'], d[workload].get('measured_at_commit') or d.get('measured_at_commit')
    except Exception:
        return None, None


def make_clusters(ds, threshold, protein=False):
    """Clusters + consensus rows of the command line's default route (run_apples.py: max-diameter
    clusters at 1.2 x -f, apples/Reference.py:87; the consensus over the alphabet of apples/PoolRepresentativeWorker.py:33-58)."""
    from apples_amd import treecluster
    from apples_amd.fasta import Alignment
    from apples_amd.reference import ReducedReference
    ref = ReducedReference(Alignment(ds.ref_names, ds.ref_seqs), bool(protein), treecluster.grouped(ds.tree, threshold * 1.2))
    return ref.cluster_arrays()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='c3', choices=sorted(WORKLOADS))
    ap.add_argument('--scaling', default='', choices=['', 'weak', 'strong'],
                    help="strong: the workload's queries are split over the ranks (default for c3 = BASELINE config 3); "
                         'weak: every rank places a block of the workload\'s size (default for the other workloads)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--queries', type=int, default=0, help='override the number of queries (per GPU when weak)')
    ap.add_argument('--gather', default='rccl', choices=['torch', 'rccl'],
                    help='the end-of-run gather for N > 1: ctypes on librccl.so without PyTorch (apples_amd/rccl.py, default) '
                         'or torch.distributed (backend nccl = RCCL)')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the untimed extras of the N = 1 line (HBM point, strong-scaling proxy, clustered route)')
    ap.add_argument('--timed', default='', choices=['', 'host', 'resident'],
                    help='what a step covers: host = host buffers in, placements in host memory (default for alignment '
                         'workloads); resident = inputs already uploaded (default for c5: a 200 k-column table block is '
                         'gigabytes over PCIe, which is the copy and not the path)')
    args = ap.parse_args()
    if not args.scaling:
        args.scaling = 'strong' if args.workload == 'c3' else 'weak'

    if args.gpus > 1 and 'RANK' not in os.environ:
        # the parent of the rank processes: starts them before anything here touches a device and never does itself
        from apples_amd.launcher import launch
        sys.exit(launch(args.gpus, os.path.abspath(__file__), sys.argv[1:]))

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and world > 1:
        args.gpus = world
    # APPLES_BENCH_FORCE_DIST=1 (tests): take the multi-rank code path -- process group, gather,
    # max over ranks -- even with a single rank, so that one GPU is enough to exercise it
    use_dist = world > 1 or bool(os.environ.get('APPLES_BENCH_FORCE_DIST'))
    dist = torch = comm = None
    if use_dist and args.gather == 'rccl':
        from apples_amd.rccl import Comm
        comm = Comm(rank, world, local_rank)
    elif use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    from apples_amd import synth
    from apples_amd.distributed import shard_bounds
    from apples_amd.engine import Engine, PLACEMENT_DTYPE

    n_leaves, L, Q, protein, method, thr = WORKLOADS[args.workload]
    if args.queries:
        Q = args.queries
    if os.environ.get('APPLES_BENCH_LEAVES'):  # experiment only (never the reported line): another backbone size for the workload
        n_leaves = int(os.environ['APPLES_BENCH_LEAVES'])
    table = args.workload == 'c5'
    clustered = args.workload.endswith('-clustered')
    strong = args.scaling == 'strong'
    # every rank holds the same backbone + reference.  weak: rank-specific blocks of Q queries (own seed);
    # strong: one set of Q queries, rank r takes the contiguous block shard_bounds gives it
    ds = synth.make_dataset(n_leaves, L if not table else 4, Q, protein=protein, seed_query=3 if strong else 3 + rank)
    lo, hi = shard_bounds(Q, world)[rank] if strong else (0, Q)
    if os.environ.get('APPLES_BENCH_SHARD'):
        # experiment only (never the reported line): this process places shard k of the 8-GPU job's split of the workload's query set
        ds = synth.make_dataset(n_leaves, L if not table else 4, WORKLOADS[args.workload][2], protein=protein, seed_query=3)
        lo, hi = shard_bounds(WORKLOADS[args.workload][2], 8)[int(os.environ['APPLES_BENCH_SHARD'])]
        Q = hi - lo
    queries = np.ascontiguousarray(ds.query_seqs[lo:hi])
    if os.environ.get('APPLES_BENCH_TREE_ORDER'):
        # experiment only (never the reported line): the block's queries in the tree order of their true sister leaves,
        # to measure what locality between neighbouring queries is worth to the selection and the sweep
        queries = np.ascontiguousarray(queries[np.argsort(ds.query_leaf[lo:hi], kind='stable')])
    sizes = [(b - a) for a, b in shard_bounds(Q, world)] if strong else [Q] * world
    nodes = np.array([ds.tree.name_to_node[n] for n in ds.ref_names], np.int32)
    D = None
    if table:
        # noisy true path distances, generated in binary (never as text)
        index = synth.TreeIndex(ds.tree)
        D = synth.fast_distance_rows(ds.tree, index, ds.query_leaf, ds.query_pendant, list(range(lo, hi)),
                                     seed_noise=7 if strong else 7 + rank)
        eng = Engine(ds.tree, None, method=method, criterion='MLSE', threshold=thr, baseobs=25, device=local_rank)
    else:
        eng = Engine(ds.tree, ds.ref_seqs, nodes, clusters=make_clusters(ds, thr, protein) if clustered else None, protein=protein,
                     method=method, criterion='MLSE', threshold=thr, baseobs=25, overlap=0.001, device=local_rank)
    nq = hi - lo
    timed = args.timed or ('resident' if table else 'host')
    res_handle = None
    if timed == 'resident':
        res_handle, _ = eng.upload_table(D, nodes) if table else eng.upload_queries(queries)

    class _DevArray:  # zero-copy view of the device-resident placement structs
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {'shape': (nbytes,), 'typestr': '|u1', 'data': (ptr, False), 'version': 2}

    if use_dist and comm is None:
        from apples_amd.distributed import gather_bytes

    def step():
        """host buffer -> placements in host memory (on rank 0 for the whole job when N > 1); with --timed resident
        the inputs are on the device already and only the placements travel."""
        if not use_dist:
            if res_handle is not None:
                eng.place_resident(res_handle)
                return eng.fetch(res_handle, nq)
            return eng.place_distances(D, nodes) if table else eng.place_sequences(queries)
        if res_handle is not None:
            h, n = res_handle, nq
            eng.place_resident(h)
        elif table:
            h, n = eng.upload_table(D, nodes)
            eng.place_resident(h)
        else:
            h, n = eng.place_sequences_streamed(queries)
        # the end-of-run gather over RCCL/xGMI (replaces starmap's pickle return), straight from the
        # device-resident structs; rank 0 then brings the whole job's placements to the host
        out = None
        if comm is not None:
            raw = comm.gather_to_host(eng.placements_device_ptr(h), [s * 40 for s in sizes])
            if rank == 0:
                out = np.frombuffer(raw, dtype=PLACEMENT_DTYPE)
        else:
            res = torch.as_tensor(_DevArray(eng.placements_device_ptr(h), n * 40), device='cuda')
            parts = gather_bytes(res, rank, world, dist, [s * 40 for s in sizes])
            if rank == 0:
                out = np.frombuffer(torch.cat(parts).cpu().numpy().tobytes(), dtype=PLACEMENT_DTYPE)
        if res_handle is None:
            eng.free_queries(h)
        return out

    def sync():
        if comm is not None:
            comm.barrier()
        elif use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    phases = {'dist_ms': 0.0, 'select_ms': 0.0, 'sweep_ms': 0.0, 'filter_ms': 0.0, 'blocks_ms': 0.0}
    launches = 0
    out = None
    for _ in range(args.steps):
        out = step()
        t = eng.timing()
        for k in phases:
            phases[k] += t[k]
        launches += t['dist_launches']
    sync()
    dt = time.perf_counter() - t0
    if comm is not None:
        dt = comm.max_over_ranks(dt)
    elif use_dist:
        tmax = torch.tensor([dt], device='cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # untimed: the resident form (query block / table already uploaded and packed; device time only)
    if res_handle is not None:
        handle = res_handle
    elif table:
        handle, _ = eng.upload_table(D, nodes)
    else:
        handle, _ = eng.upload_queries(queries)
    eng.place_resident(handle)
    sync()
    t1 = time.perf_counter()
    res_phases = {'dist_ms': 0.0, 'select_ms': 0.0, 'sweep_ms': 0.0}
    for _ in range(args.steps):
        eng.place_resident(handle)
        t = eng.timing()
        for k in res_phases:
            res_phases[k] += t[k]
    dt_res = time.perf_counter() - t1
    final_line = None
    mine = eng.fetch(handle, nq)
    eng.free_queries(handle)
    if rank == 0:
        mine0 = out[:nq]
        if mine0.tobytes() != mine.tobytes():
            raise SystemExit('bench: the host-buffer pass and the resident pass disagree')
        total_q = int(sum(sizes))
        ms_per_step = dt / args.steps * 1e3
        value = total_q / (dt / args.steps)
        rows = eng.n_rows if not table else n_leaves
        placed = mine['n_valid'] > 0
        mean_v = float(np.mean(mine['n_valid'][placed] + 1)) if placed.any() else 0.0
        per_step = {k: v / args.steps for k, v in phases.items() if k != 'filter_ms' and (k != 'blocks_ms' or v > 0)}
        info = eng.describe()
        roofline = roofline_of(args.workload, nq, rows, L, protein, table, per_step, launches / args.steps, mine, info,
                               filter_ms=phases['filter_ms'] / args.steps)
        stream = proxy = None
        if world == 1 and not table and not protein and not clustered:
            stream = distance_stream_point(eng, ds, L)
        extras = world == 1 and not args.no_extras and not use_dist
        if extras and args.workload == 'c3':
            proxy = strong_scaling_proxy(eng, queries, ms_per_step, device=local_rank, full=mine)
        cpu = None
        if world == 1 and not args.no_cpu and not clustered:
            cpu = cpu_baseline_table(ds, D, method, thr) if table else cpu_baseline(ds, protein, method, thr)
        res_ms = dt_res / args.steps * 1e3
        line = {
            'metric': 'query placements/sec (whole node)', 'value': value, 'unit': 'queries/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': ('%s: synthetic %d-leaf backbone, -d distance-table input (noisy path distances, binary), '
                                    '%d queries%s, %s, -f %.1f -b 25' % (args.workload, n_leaves, Q, '' if strong else ' per GPU', method, thr)) if table
                       else '%s: synthetic %d-leaf backbone, L=%d %s, %d queries%s, %s/%s, -f %.1f -b 25, %s'
                       % (args.workload, n_leaves, L, 'aa' if protein else 'nt', Q, ' in all' if strong else ' per GPU',
                          method, 'scoredist' if protein else 'JC69', thr,
                          'max-diameter clusters at 1.2 x -f with consensus representatives (%d representatives)' % info['n_reps']
                          if clustered else 'all-singleton clusters'),
                       'timed': 'host byte arrays -> placements in host memory (upload, packing, kernels, copy back)' if timed == 'host'
                       else 'inputs resident in HBM -> placements in host memory (kernels, copy back)',
                       'n_ref': n_leaves, 'L': L, 'queries_this_rank': nq, 'queries_total': total_q, 'method': method,
                       'mean_observed': float(np.mean(mine['n_obs'])), 'mean_swept_nodes': mean_v,
                       'placed': int(placed.sum()), 'parallelism': 'query-sharded x%d' % world,
                       # every placement struct of the job as rank 0 holds them after the gather (strong scaling: the same bytes at any N)
                       'placements_crc32': zlib.crc32(np.ascontiguousarray(out).tobytes()),
                       'gather': ('librccl via ctypes (no PyTorch)' if comm is not None else 'torch.distributed nccl') if use_dist else None},
            'roofline': roofline,
            'resident': {'value': world * nq / (dt_res / args.steps), 'ms_per_step': res_ms, 'unit': 'queries/s',
                         'per_kernel_ms_per_step': {k: v / args.steps for k, v in res_phases.items()},
                         'note': 'same pass, query block already uploaded and packed (rank 0, device time only)'},
            'distance_kernel_stream': stream,
            'cpu_baseline': cpu,
        }
        if cpu:
            line['speedup_vs_cpu_baseline'] = value / cpu['value']
        if proxy:
            line['strong_scaling_proxy'] = proxy
    eng.close()
    if rank == 0:
        if extras and args.workload == 'c3':
            line['clustered'] = clustered_leg(ds, nodes, queries, thr, method, local_rank)
        if extras and not table and not protein:
            line['roofline_hbm_point'] = hbm_point(ds.ref_seqs, queries, local_rank)
        if extras and args.workload == 'c3':
            # the other BASELINE configs, so that the driver's own run sees them (config 5 on config 3's tree: the same backbone)
            line['other_workloads'] = {'c2': other_workload('c2', local_rank), 'c4': other_workload('c4', local_rank),
                                       # config 4's inputs through the default route of `-p`: clusters at 1.2 x -f, consensus representatives
                                       'c4_clustered': other_workload('c4-clustered', local_rank),
                                       'c5': other_workload('c5', local_rank, ds=ds),
                                       # config 5 as one rank of its 8-GPU job holds it: 12 500 of the 100 000 rows (20 GB) resident
                                       'c5_shard_12500_rows': other_workload('c5', local_rank, ds=ds, queries=12500),
                                       # SURVEY 8d's stress variant at config 3's shape: -f 1e9, every leaf observed, V = 2 N - 2 swept
                                       # nodes per query (133 MB of algorithmic sweep bytes each), on the first 2 048 queries
                                       'c3_all_observed_2048_queries': other_workload('c3-all', local_rank, ds=ds)}
            # config 3's size in the shapes real inputs have (variant_dataset), each with the route that served it, singleton
            # clusters and the command line's default route
            shapes = {}
            for v in ('unrooted', 'polytomies', 'deep', 'dots', 'L4000'):
                dv = variant_dataset(ds, v, n_leaves, L, 25000 if v == 'L4000' else Q, protein)
                shapes['c3-' + v] = other_workload('c3', local_rank, ds=dv, variant=v, prepared=True, queries=25000 if v == 'L4000' else 0)
                if v in ('unrooted', 'polytomies', 'deep'):
                    shapes['c3-%s-clustered' % v] = other_workload('c3-clustered', local_rank, ds=dv, variant=v, prepared=True)
                del dv
            shapes['c3-clustered-300k'] = other_workload('c3-clustered', local_rank, variant='300k')
            line['other_shapes'] = shapes
        final_line = json.dumps(line)
    if comm is not None:
        comm.barrier()
        comm.close()
    elif use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line comes last: after the communicator is gone and whatever the collective library
        # left in the C stdio buffer (its version banner) has been flushed
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(final_line, flush=True)


if __name__ == '__main__':
    main()
