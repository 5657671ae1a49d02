#!/usr/bin/env python3
"""Benchmark of the APPLES per-query hot path on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|small] [--no-cpu]

A step = one pass of the hot path (distance vector -> observed-set selection -> least-squares
sweep -> best edge) over one block of synthetic queries whose packed form, together with the
packed reference alignment and the tree, is already resident in HBM.  value = query
placements/s for the whole job.  For N > 1 (launched by torch.distributed.run, one rank per GPU)
every rank places its own shard of the same size (weak scaling, no collective in the data path)
and the placements are gathered to rank 0 with one RCCL gather inside the timed region.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, algorithmic bytes / HIP-event
time), `distance_kernel_stream` (BASELINE.json's second figure: the distance kernel
at one query per pass over the packed reference, bytes moved / HIP-event time against the 8 TB/s
peak, measured after the timed region) and `cpu_baseline` (the CPU restatement of the reference
path -- per-query numpy/Python worker under a fork pool, as run_apples.py:101-102 -- timed on this
host's cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (leaves, L, queries per GPU, protein, method, threshold)   -- BASELINE.json configs
    'c2': (10000, 1000, 10000, False, 'OLS', 0.2),
    'c3': (200000, 1000, 100000, False, 'OLS', 0.2),
    'c4': (50000, 500, 50000, True, 'FM', 0.2),
    'small': (2000, 500, 2048, False, 'OLS', 0.2),
    # C5: distance-table input (-d), least-squares-only path; L is irrelevant (no alignment)
    'c5': (200000, 0, 100000, False, 'BME', 0.2),
}
MFMA_F4_PEAK_TOPS = 10000.0  # dense fp4 peak at the nominal clock, /opt/skills/guides/MI355X_MICROARCH.md, matrix cores table
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def cpu_baseline(ds, protein, method, threshold, target_cpu_seconds=20.0):
    """Time the oracle's pool driver on a bounded, seeded sample of the same queries."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import apples_oracle as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    kw = dict(protein=protein, method=method, criterion='MLSE', threshold=threshold, baseobs=25, overlap=0.001)
    t0 = time.time()
    orc.run_pool(ds.tree, ds.ref_names, ds.ref_seqs, ds.query_names[:1], ds.query_seqs[:1], threads=1, **kw)
    t1 = max(time.time() - t0, 1e-3)
    # every core gets 8 queries; the pool is started and warmed before the clock starts (start-up,
    # which the reference's own "Processed all queries" timer would include, is reported apart)
    n = int(min(len(ds.query_names), max(8 * cores, min(16 * cores, target_cpu_seconds / t1))))
    dt, startup, _ = orc.time_pool(ds.tree, ds.ref_names, ds.ref_seqs, ds.query_names[:n], ds.query_seqs[:n], cores, **kw)
    return {'value': n / dt, 'unit': 'queries/s', 'cores': cores, 'kind': 'port',
            'sample': 'first %d of the %d synthetic queries on a warmed %d-process fork pool: %.2f s steady state '
                      '(pool start-up %.1f s not counted; %.3f s for one query on one core)'
                      % (n, len(ds.query_names), cores, dt, startup, t1)}


def cpu_baseline_table(ds, D, method, threshold, target_cpu_seconds=20.0):
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import apples_oracle as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    t0 = time.time()
    orc.time_pool_table(ds.tree, ds.ref_names, ds.query_names[:1], D[:1], 1, method=method, threshold=threshold)
    t1 = max(time.time() - t0, 1e-3)
    n = int(min(len(D), max(2 * cores, min(8 * cores, target_cpu_seconds / t1))))
    dt, startup, _ = orc.time_pool_table(ds.tree, ds.ref_names, ds.query_names[:n], D[:n], cores, method=method,
                                         threshold=threshold)
    return {'value': n / dt, 'unit': 'queries/s', 'cores': cores, 'kind': 'port',
            'sample': 'first %d table rows on a warmed %d-process fork pool: %.2f s steady state (start-up %.1f s not '
                      'counted; %.3f s for one query on one core)' % (n, cores, dt, startup, t1)}


def distance_stream_point(eng, ds, L):
    """BASELINE.json's second figure, measured live: the distance kernel at query tile T = 1 (SURVEY
    8d's roofline point: the packed reference streamed once per query, full fp64 rows out), on up to
    256 of the workload's queries, outside the timed region.  `delivered` counts the bytes the kernel
    moves (packed reference + 8 B per pair); whether they come from HBM or from the caches depends
    on the reference's size, which is reported beside it."""
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
            'served_from': 'L2 / Infinity Cache (reference smaller than the 256 MiB cache)'
                           if info['packed_bytes'] < (256 << 20) else 'HBM'}


def load_traffic(workload, kernel):
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 PMC passes
    (profiles/pmc_summary.json, written by scripts/pmc_to_traffic.py from separate --pmc runs of
    this same command; FETCH_SIZE/WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes)."""
    path = os.path.join(ROOT, 'profiles', 'pmc_summary.json')
    try:
        with open(path) as f:
            d = json.load(f)
        return d[workload][kernel]['hbm_bytes_per_launch']
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--queries', type=int, default=0, help='override queries per GPU')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and world > 1:
        args.gpus = world
    # APPLES_BENCH_FORCE_DIST=1 (tests): take the multi-rank code path -- process group, gather,
    # max over ranks -- even with a single rank, so that one GPU is enough to exercise it
    use_dist = world > 1 or bool(os.environ.get('APPLES_BENCH_FORCE_DIST'))
    dist = torch = None
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    from apples_amd import synth
    from apples_amd.engine import Engine

    n_leaves, L, Q, protein, method, thr = WORKLOADS[args.workload]
    if args.queries:
        Q = args.queries
    # every rank holds the same backbone + reference; queries are rank-specific shards
    table = args.workload == 'c5'
    ds = synth.make_dataset(n_leaves, L if not table else 4, Q, protein=protein, seed_query=3 + rank)
    nodes = np.array([ds.tree.name_to_node[n] for n in ds.ref_names], np.int32)
    D = None
    if table:
        # noisy true path distances, generated in binary (never as text)
        index = synth.TreeIndex(ds.tree)
        D = synth.fast_distance_rows(ds.tree, index, ds.query_leaf, ds.query_pendant, list(range(Q)), seed_noise=7 + rank)
        eng = Engine(ds.tree, None, method=method, criterion='MLSE', threshold=thr, baseobs=25, device=local_rank)
        handle, nq = eng.upload_table(D, nodes)
    else:
        eng = Engine(ds.tree, ds.ref_seqs, nodes, protein=protein, method=method, criterion='MLSE', threshold=thr,
                     baseobs=25, overlap=0.001, device=local_rank)
        handle, nq = eng.upload_queries(ds.query_seqs)

    if use_dist:
        class _DevArray:  # zero-copy view of the device-resident placement structs
            def __init__(self, ptr, nbytes):
                self.__cuda_array_interface__ = {'shape': (nbytes,), 'typestr': '|u1', 'data': (ptr, False), 'version': 2}
        res = torch.as_tensor(_DevArray(eng.placements_device_ptr(handle), nq * 40), device='cuda')
        from apples_amd.distributed import gather_bytes

    def step():
        eng.place_resident(handle)  # returns after the stream has drained
        if use_dist:
            # the end-of-run gather over RCCL/xGMI (replaces starmap's pickle return)
            gather_bytes(res, rank, world, dist)

    def sync():
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    phases = {'dist_ms': 0.0, 'select_ms': 0.0, 'sweep_ms': 0.0}
    launches = 0
    for _ in range(args.steps):
        step()
        t = eng.timing()
        for k in phases:
            phases[k] += t[k]
        launches += t['dist_launches']
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], device='cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    out = eng.fetch(handle, nq)
    if use_dist:  # untimed: what the gather delivers for this rank is what the device holds
        parts = gather_bytes(res, rank, world, dist)
        torch.cuda.synchronize()
        if rank == 0 and parts[0].cpu().numpy().tobytes() != out.tobytes():
            raise SystemExit('bench: gathered placements differ from the device buffer')
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * nq / (dt / args.steps)
        rows = eng.n_rows if not table else n_leaves
        placed = out['n_valid'] > 0
        mean_v = float(np.mean(out['n_valid'][placed] + 1)) if placed.any() else 0.0
        per_step = {k: v / args.steps for k, v in phases.items()}
        # algorithmic bytes per step: distance N_rows*(L+8)+L per query; sweep 332*V per query
        dist_bytes = nq * (rows * (L + 8) + L)
        sweep_bytes = 332.0 * float(np.sum(out['n_valid'][placed] + 1))
        kernels = {'lsq_sweep': (sweep_bytes, per_step['sweep_ms'])}
        if table:  # the -d filter reads every table value once
            kernels['table_select'] = (nq * rows * 8.0, per_step['select_ms'])
        else:
            kernels['jc69_distance' if not protein else 'scoredist_distance'] = (dist_bytes, per_step['dist_ms'])
        dom = max(kernels, key=lambda k: kernels[k][1])
        n_launch = max(launches / args.steps, 1) if dom != 'lsq_sweep' else max(launches / args.steps, 1)
        achieved = kernels[dom][0] / (kernels[dom][1] * 1e-3) / 1e9 if kernels[dom][1] > 0 else 0.0
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': achieved / HBM_PEAK_GBS, 'traffic': load_traffic(args.workload, dom),
                    'launches_per_step': n_launch, 'avg_launch_ms': kernels[dom][1] / n_launch,
                    'algorithmic_bytes_per_launch': kernels[dom][0] / n_launch,
                    'per_kernel_ms_per_step': per_step,
                    'all_kernels_GBps': {k: (v[0] / (v[1] * 1e-3) / 1e9 if v[1] > 0 else 0.0) for k, v in kernels.items()}}
        if dom == 'jc69_distance' and eng.describe().get('code_planes') == 2 and not os.environ.get('APPLES_NO_DIST_MFMA'):
            # the tiled pair counts run on the matrix cores (fp4 operands, 4 MACs per site and pair:
            # DESIGN.md section 4): price them against the dense fp4 MFMA peak, 2 ops per MAC
            ops = 2.0 * 4.0 * nq * rows * 32.0 * ((L + 31) // 32)
            tops = ops / (kernels[dom][1] * 1e-3) / 1e12
            roofline.update({'bound': 'mfma', 'achieved': tops, 'peak': MFMA_F4_PEAK_TOPS, 'unit': 'TFLOP/s',
                             'frac': tops / MFMA_F4_PEAK_TOPS, 'algorithmic_ops_per_launch': ops / n_launch,
                             'hbm_algorithmic_GBps': achieved})
        stream = None
        if world == 1 and not table and not protein:
            stream = distance_stream_point(eng, ds, L)
        cpu = None
        if world == 1 and not args.no_cpu:
            cpu = cpu_baseline_table(ds, D, method, thr) if table else cpu_baseline(ds, protein, method, thr)
        line = {
            'metric': 'query placements/sec (whole node)', 'value': value, 'unit': 'queries/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': ('%s: synthetic %d-leaf backbone, -d distance-table input (noisy path distances, binary), '
                                    '%d queries per GPU, %s, -f %.1f -b 25' % (args.workload, n_leaves, nq, method, thr)) if table
                       else '%s: synthetic %d-leaf backbone, L=%d %s, %d queries per GPU, %s/%s, -f %.1f -b 25, '
                       'all-singleton clusters' % (args.workload, n_leaves, L, 'aa' if protein else 'nt', nq,
                                                  method, 'scoredist' if protein else 'JC69', thr),
                       'n_ref': n_leaves, 'L': L, 'queries_per_gpu': nq, 'method': method,
                       'mean_observed': float(np.mean(out['n_obs'])), 'mean_swept_nodes': mean_v,
                       'placed': int(placed.sum()), 'parallelism': 'query-sharded x%d' % world},
            'roofline': roofline,
            'distance_kernel_stream': stream,
            'cpu_baseline': cpu,
        }
        if cpu:
            line['speedup_vs_cpu_baseline'] = value / cpu['value']
        print(json.dumps(line), flush=True)
    eng.free_queries(handle)
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
