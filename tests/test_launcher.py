"""bench.py --gpus N starts its own ranks (apples_amd/launcher.py; in place of the reference's fork pool,
run_apples.py:93-102).  Driven here on the CPU with a stub rank: the parent must never need a device."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')

STUB = r'''
import json, os, sys, time
rank = int(os.environ['RANK'])
mode = os.environ.get('STUB_MODE', 'ok')
if mode == 'fail' and rank == 1:
    sys.exit(7)
if mode == 'fail' and rank != 1:
    time.sleep(60)       # must be stopped by the launcher, not run to its end
if mode == 'hang':
    open(os.environ['STUB_PIDS'], 'a').write('%d\n' % os.getpid())
    time.sleep(120)      # the launcher is terminated from outside: its ranks must go with it
print(json.dumps({'rank': rank, 'local_rank': int(os.environ['LOCAL_RANK']), 'world': int(os.environ['WORLD_SIZE']),
                  'addr': os.environ['MASTER_ADDR'], 'port': int(os.environ['MASTER_PORT']), 'argv': sys.argv[1:],
                  'ipc': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}), flush=True)
'''


def _run(tmp_path, gpus, devices, mode='ok', extra=()):
    stub = tmp_path / 'stub_rank.py'
    stub.write_text(STUB)
    env = dict(os.environ)
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    env.update({'APPLES_LAUNCH_DEVICE_COUNT': str(devices), 'APPLES_LAUNCH_RANK_CMD': json.dumps([sys.executable, str(stub)]),
                'STUB_MODE': mode})
    return subprocess.run([sys.executable, BENCH, '--gpus', str(gpus), '--steps', '2', '--warmup', '1'] + list(extra),
                          env=env, capture_output=True, text=True, timeout=120)


def test_launcher_starts_one_rank_per_gpu_and_relays_rank0(tmp_path):
    r = _run(tmp_path, 4, 8)
    assert r.returncode == 0, r.stderr
    out = [json.loads(x) for x in r.stdout.splitlines() if x.strip()]
    assert len(out) == 1 and out[0]['rank'] == 0            # ONE line on stdout: rank 0's
    others = [json.loads(x) for x in r.stderr.splitlines() if x.startswith('{')]
    assert sorted(o['rank'] for o in others) == [1, 2, 3]
    for o in out + others:
        assert o['world'] == 4 and o['local_rank'] == o['rank'] and o['addr'] == '127.0.0.1' and o['ipc'] == '0'
        assert o['port'] == out[0]['port']
        assert o['argv'] == ['--gpus', '4', '--steps', '2', '--warmup', '1']


def test_launcher_refuses_more_ranks_than_devices(tmp_path):
    r = _run(tmp_path, 2, 1)
    assert r.returncode != 0
    assert '2 ranks requested, 1 device visible' in r.stderr
    assert r.stdout.strip() == ''


def test_launcher_failed_rank_stops_the_others(tmp_path):
    import time
    t0 = time.time()
    r = _run(tmp_path, 3, 3, mode='fail')
    assert r.returncode == 7
    assert 'rank 1 ended with status 7' in r.stderr
    assert time.time() - t0 < 30


def test_terminating_the_launcher_takes_the_ranks_down(tmp_path):
    """`timeout N python bench.py --gpus 8` ends the parent with SIGTERM: the ranks must not stay behind with their GPUs."""
    import signal
    import time
    stub = tmp_path / 'stub_rank.py'
    stub.write_text(STUB)
    pids = tmp_path / 'pids'
    env = dict(os.environ)
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    env.update({'APPLES_LAUNCH_DEVICE_COUNT': '3', 'APPLES_LAUNCH_RANK_CMD': json.dumps([sys.executable, str(stub)]),
                'STUB_MODE': 'hang', 'STUB_PIDS': str(pids)})
    p = subprocess.Popen([sys.executable, BENCH, '--gpus', '3'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    t0 = time.time()
    while time.time() - t0 < 60 and (not pids.exists() or len(pids.read_text().split()) < 3):
        time.sleep(0.1)
    ranks = [int(x) for x in pids.read_text().split()]
    assert len(ranks) == 3
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=30)
    assert p.returncode == 128 + signal.SIGTERM
    t1 = time.time()
    alive = ranks
    while alive and time.time() - t1 < 15:
        alive = [r for r in alive if os.path.exists('/proc/%d' % r) and open('/proc/%d/stat' % r).read().split()[2] != 'Z']
        time.sleep(0.1)
    assert not alive, 'ranks left behind: %s' % alive


def test_defaults_strong_for_c3_and_torch_free_gather():
    src = open(BENCH).read()
    assert "default='rccl'" in src and "'strong' if args.workload == 'c3'" in src


def test_parent_does_not_import_the_engine_or_torch(tmp_path):
    """The launching parent must stay free of GPU state: with the rank command stubbed, a bench.py whose engine
    import would fail (no library path) still launches."""
    stub = tmp_path / 'stub_rank.py'
    stub.write_text(STUB)
    env = dict(os.environ)
    env.pop('RANK', None)
    env.update({'APPLES_LAUNCH_DEVICE_COUNT': '2', 'APPLES_LAUNCH_RANK_CMD': json.dumps([sys.executable, str(stub)])})
    code = ('import sys, runpy\n'
            'sys.argv = [%r, "--gpus", "2"]\n'
            'import builtins\n'
            'real = builtins.__import__\n'
            'def guard(name, *a, **k):\n'
            '    if name.split(".")[0] == "torch" or name in ("apples_amd.engine", "apples_amd.rccl"):\n'
            '        raise AssertionError("parent imported " + name)\n'
            '    return real(name, *a, **k)\n'
            'builtins.__import__ = guard\n'
            'runpy.run_path(%r, run_name="__main__")\n' % (BENCH, BENCH))
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert 'parent imported' not in r.stderr


def _side_rank(rank, world, port, q):
    from apples_amd.rccl import SideChannel
    try:
        ch = SideChannel(rank, world, blob=(b'id' * 64 if rank == 0 else None), addr='127.0.0.1', port=port, timeout=30.0)
        m = ch.max_over_ranks(10.0 + rank)
        m2 = ch.max_over_ranks(-float(rank))
        q.put((rank, bytes(ch.blob), m, m2))
        ch.close()
    except Exception as e:  # noqa: BLE001 (reported to the parent)
        q.put((rank, repr(e), None, None))


def test_side_channel_rendezvous_skips_a_foreign_listener():
    """apples_amd/rccl.py's TCP side channel (the communicator id, the barrier and the max-over-ranks travel over it) with
    three processes on the CPU; the first port of its range is held by somebody else's listener, which answers with
    something that is not this job's greeting: every rank moves on to the next port."""
    import multiprocessing as mp
    import socket
    import threading
    from apples_amd.launcher import free_port
    port = free_port()
    foreign = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    foreign.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    foreign.bind(('127.0.0.1', port))
    foreign.listen(8)
    stop = threading.Event()

    def serve():
        foreign.settimeout(0.2)
        while not stop.is_set():
            try:
                c, _ = foreign.accept()
            except OSError:
                continue
            try:
                c.sendall(b'HTTP/1.1 400 Bad Request\r\n\r\n' + b'x' * 64)
            finally:
                c.close()
    th = threading.Thread(target=serve, daemon=True)
    th.start()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_side_rank, args=(r, world, port, q)) for r in (2, 1, 0)]  # (rank 0 last: the others wait for it)
    try:
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=60) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()
        stop.set()
        th.join(timeout=2)
        foreign.close()
    assert [g[0] for g in got] == [0, 1, 2]
    for r, blob, m, m2 in got:
        assert blob == b'id' * 64, (r, blob)
        assert m == 12.0 and m2 == 0.0


def _side_rank_late(rank, world, port, q, delay):
    import time
    time.sleep(delay)
    _side_rank(rank, world, port, q)


def test_side_channel_rendezvous_with_ranks_arriving_seconds_apart():
    """Ranks that reach rank 0 more than the client's 3-second greeting timeout apart (round 4's advisor finding: rank 0 used to
    echo the greeting only once everybody was in, an early rank gave up on its connection and its retry was refused as a
    duplicate).  The echo now goes out as each rank is accepted; a rank that does come back replaces its dead connection."""
    import multiprocessing as mp
    from apples_amd.launcher import free_port
    port = free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_side_rank_late, args=(r, world, port, q, d)) for r, d in ((0, 0.0), (1, 0.0), (2, 4.5))]
    try:
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=60) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()
    assert [g[0] for g in got] == [0, 1, 2]
    for r, blob, m, m2 in got:
        assert blob == b'id' * 64, (r, blob)
        assert m == 12.0 and m2 == 0.0
