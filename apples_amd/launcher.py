"""One process per GPU, started by a parent that never touches a device.

The reference spreads its queries over a fork pool, `mp.Pool(num_thread).starmap(runquery, queries)`
(run_apples.py:93-102).  Here the pool is N rank processes, one per GPU of the node: the parent counts the
devices in a throw-away child (so that it never initialises HIP itself -- a process that has may not start
others safely on this pool, and nothing is ever re-exec'ed), refuses loudly when fewer devices than ranks are
visible, spawns N fresh interpreters with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set,
relays rank 0's standard output (the one JSON line) and waits.  A rank that fails takes the others down
with it (each child is ended by its own PID) and its exit code becomes the launcher's.

    python bench.py --gpus N ...        (no RANK in the environment: bench.py calls launch())

Test hooks (tests/test_launcher.py drives the launcher on the CPU with a stub rank):
APPLES_LAUNCH_DEVICE_COUNT = pretend this many devices are visible; APPLES_LAUNCH_RANK_CMD = JSON list that
replaces `[python, script]` as the rank's command line."""
import json
import os
import signal
import socket
import subprocess
import sys
import time

_COUNT_SNIPPET = (
    'import ctypes, sys\n'
    'n = ctypes.c_int(0)\n'
    'try:\n'
    '    hip = ctypes.CDLL("libamdhip64.so")\n'
    'except OSError:\n'
    '    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")\n'
    'rc = hip.hipGetDeviceCount(ctypes.byref(n))\n'
    'print(n.value if rc == 0 else 0)\n')


def visible_devices():
    """Number of HIP devices, asked of a child process: the caller stays free of any GPU state."""
    forced = os.environ.get('APPLES_LAUNCH_DEVICE_COUNT')
    if forced is not None:
        return int(forced)
    try:
        out = subprocess.run([sys.executable, '-c', _COUNT_SNIPPET], capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 and out.stdout.strip() else 0
    except (OSError, ValueError, subprocess.TimeoutExpired):
        return 0


def free_port():
    """A TCP port pair (p, p + 1) free on 127.0.0.1: MASTER_PORT and the side channel of apples_amd/rccl.py."""
    for _ in range(64):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(('127.0.0.1', 0))
            p = s.getsockname()[1]
        if p >= 65535:
            continue
        try:
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s2:
                s2.bind(('127.0.0.1', p + 1))
            return p
        except OSError:
            continue
    raise RuntimeError('no free port pair on 127.0.0.1')


def rank_environment(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': str(world), 'LOCAL_WORLD_SIZE': str(world),
                'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port),
                # names the job in the side channel's greeting (apples_amd/rccl.py): two jobs on one port pair stay apart
                'APPLES_JOB_NONCE': env.get('APPLES_JOB_NONCE') or str((os.getpid() << 20) ^ (time.time_ns() & 0xfffffffffff)),
                # dmabuf IPC: what the host driver supports (RCCL needs it between processes)
                'HSA_ENABLE_IPC_MODE_LEGACY': env.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')})
    return env


def launch(n_ranks, script, argv, timeout=None, stdout=None, stderr=None):
    """Start n_ranks copies of `python script argv...`, one per device; returns the exit code (0 = every rank ended
    cleanly).  Rank 0 writes to `stdout` (default: ours), the other ranks' standard output goes to `stderr`."""
    n_ranks = int(n_ranks)
    have = visible_devices()
    if have < n_ranks:
        msg = 'bench: %d ranks requested, %d device%s visible: one process per GPU, no oversubscription' \
              % (n_ranks, have, '' if have == 1 else 's')
        print(msg, file=stderr or sys.stderr, flush=True)
        return 2
    cmd = json.loads(os.environ['APPLES_LAUNCH_RANK_CMD']) if os.environ.get('APPLES_LAUNCH_RANK_CMD') \
        else [sys.executable, script]
    port = free_port()
    procs = []

    # `timeout 600 python bench.py --gpus 8` and harness kills deliver SIGTERM / SIGHUP: without a handler the interpreter dies at
    # once, the `finally` below never runs and the ranks stay behind holding their GPUs (blocked in a rendezvous or a
    # collective).  The handlers turn the signal into an exception; the previous ones come back when launch() returns.
    def _stop(signum, _frame):
        raise SystemExit(128 + signum)

    previous = {}
    for sig in (signal.SIGTERM, signal.SIGHUP):
        try:
            if signal.getsignal(sig) == signal.SIG_IGN:  # `nohup python bench.py --gpus 8 &`: a hang-up stays ignored
                continue
            previous[sig] = signal.signal(sig, _stop)
        except (ValueError, OSError):  # not the main thread: the caller owns signal handling
            pass
    base = dict(os.environ)
    base['APPLES_JOB_NONCE'] = rank_environment(0, n_ranks, port)['APPLES_JOB_NONCE']  # one nonce for every rank of the job
    try:
        for r in range(n_ranks):
            procs.append(subprocess.Popen(cmd + list(argv), env=rank_environment(r, n_ranks, port, base),
                                          stdout=(stdout if r == 0 else (stderr or sys.stderr)), stderr=stderr))
        deadline = None if not timeout else time.time() + float(timeout)
        rc = 0
        live = set(range(n_ranks))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
                    print('bench: rank %d ended with status %d; stopping the other ranks' % (r, code),
                          file=stderr or sys.stderr, flush=True)
            if rc != 0 or (deadline and time.time() > deadline):
                if rc == 0:
                    rc = 124
                    print('bench: ranks still running after %.0f s; stopping them' % float(timeout),
                          file=stderr or sys.stderr, flush=True)
                break
            if live:
                time.sleep(0.05)
        return rc
    finally:
        for sig in previous:  # a second signal must not cut the clean-up short and leave ranks behind
            signal.signal(sig, signal.SIG_IGN)
        for p in procs:   # by exact PID, never by pattern
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        t_end = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for sig, h in previous.items():
            signal.signal(sig, h)
