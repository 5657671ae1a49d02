"""The end-of-run gather over RCCL/xGMI without PyTorch: ctypes on librccl.so and libamdhip64.so.

Queries are sharded over the GPUs of one node, one process per GPU, with no collective in the data path; the
40-byte placement structs are gathered to rank 0 once (``starmap``'s pickle return in the reference,
run_apples.py:101-102).  Here that gather is one grouped ``ncclSend``/``ncclRecv`` straight from the
device-resident structs (``apples_placements_device_ptr``).  The 128-byte communicator id travels from rank 0
to the others over a plain TCP socket on ``MASTER_ADDR``; the same socket carries the host-side barrier and
the max-over-ranks of the timing (a few bytes per rank: no reason to spend a collective on them).

``apples_amd.distributed`` does the same through ``torch.distributed`` (what ``bench.py`` uses by default, and
gloo in the CPU tests); this module is the torch-free form (``bench.py --gather rccl``).
"""
import ctypes as C
import os
import socket
import struct
import time

NCCL_UINT8 = 1
_PORT_SPAN = 16  # ports tried for the side channel, from MASTER_PORT + 1


class _UniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]


def _lib(name, env):
    for cand in (os.environ.get(env), name, os.path.join('/opt/rocm/lib', name)):
        if not cand:
            continue
        try:
            return C.CDLL(cand)
        except OSError:
            continue
    raise RuntimeError('%s not found (set %s)' % (name, env))


class SideChannel:
    """The TCP side channel through rank 0: carries one blob from rank 0 to every rank at start-up (the communicator id), then
    the host-side barrier and the max-over-ranks of the timing.  Plain sockets, no GPU: tests/test_launcher.py runs it with
    three processes on the CPU.

    It listens on the first free port of MASTER_PORT + 1 .. + 16 (MASTER_PORT itself belongs to whoever launched the ranks);
    a connection is one of ours only if it opens with this job's greeting, so a rank that reaches somebody else's listener
    on one of those ports moves on to the next."""

    def __init__(self, rank, world, blob=None, addr=None, port=None, timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
        base = int(port or int(os.environ.get('MASTER_PORT', '29500')) + 1)
        self.peers = []   # rank 0: sockets of ranks 1..world-1, by rank
        self.sock = None  # other ranks: socket to rank 0
        self.blob = blob
        # the greeting names the job: world size, base port and the launcher's nonce (APPLES_JOB_NONCE, apples_amd/launcher.py), so
        # that two jobs sharing MASTER_PORT and world size do not take each other's ranks
        hello = struct.pack('<8siiq', b'APPLESRV', self.world, base, int(os.environ.get('APPLES_JOB_NONCE', '0') or 0))
        if self.world == 1:
            return
        if self.rank == 0:
            srv = None
            for p in range(base, base + _PORT_SPAN):
                cand = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                cand.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    cand.bind((addr, p))
                    srv = cand
                    break
                except OSError:
                    cand.close()
            if srv is None:
                raise RuntimeError('no free port for the rendezvous in %d..%d on %s' % (base, base + _PORT_SPAN - 1, addr))
            srv.listen(self.world)
            srv.settimeout(timeout)
            got = {}
            while len(got) < self.world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(10.0)
                try:
                    if self._recv(conn, len(hello)) != hello:
                        raise RuntimeError('not a rank of this job')
                    r = struct.unpack('<i', self._recv(conn, 4))[0]
                    if not 0 < r < self.world:
                        raise RuntimeError('unexpected rank %d' % r)
                    # the echo goes out at once: the client's short timeout guards only this exchange, however far apart the
                    # ranks arrive (the blob follows once everybody is in)
                    conn.sendall(hello)
                except (RuntimeError, OSError):
                    conn.close()
                    continue
                conn.settimeout(timeout)
                if r in got:  # the same rank again: its first connection is dead (it gave up on it and came back)
                    got[r].close()
                got[r] = conn
            srv.close()
            self.peers = [got[r] for r in range(1, self.world)]
            for conn in self.peers:
                conn.sendall(struct.pack('<i', len(blob or b'')) + bytes(blob or b''))
        else:
            deadline = time.time() + timeout
            while self.sock is None:
                for p in range(base, base + _PORT_SPAN):
                    try:
                        s = socket.create_connection((addr, p), timeout=2.0)
                    except OSError:
                        continue
                    try:
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        s.settimeout(3.0)  # a foreign listener that accepts and stays silent costs seconds, not the whole deadline
                        s.sendall(hello + struct.pack('<i', self.rank))
                        if self._recv(s, len(hello)) != hello:
                            raise RuntimeError('not rank 0 of this job')
                        s.settimeout(timeout)  # ours: the blob may take as long as rank 0 needs to gather everybody
                        n = struct.unpack('<i', self._recv(s, 4))[0]
                        self.blob = self._recv(s, n) if n else b''
                        self.sock = s
                        break
                    except (RuntimeError, OSError):
                        s.close()
                if self.sock is None:
                    if time.time() > deadline:
                        raise RuntimeError('rank %d found no rank 0 on %s:%d..%d' % (self.rank, addr, base, base + _PORT_SPAN - 1))
                    time.sleep(0.1)

    @staticmethod
    def _recv(conn, n):
        buf = b''
        while len(buf) < n:
            part = conn.recv(n - len(buf))
            if not part:
                raise RuntimeError('rendezvous peer closed the connection')
            buf += part
        return buf

    def max_over_ranks(self, value):
        """max of one float over the ranks (every rank gets it); doubles as a barrier."""
        if self.world == 1:
            return float(value)
        if self.rank == 0:
            vals = [float(value)] + [struct.unpack('<d', self._recv(c, 8))[0] for c in self.peers]
            m = max(vals)
            for c in self.peers:
                c.sendall(struct.pack('<d', m))
            return m
        self.sock.sendall(struct.pack('<d', float(value)))
        return struct.unpack('<d', self._recv(self.sock, 8))[0]

    def close(self):
        for c in self.peers:
            c.close()
        self.peers = []
        if self.sock:
            self.sock.close()
            self.sock = None


class _StdoutToStderr:
    """While active, file descriptor 1 points at file descriptor 2: the collective library writes a version banner to the C
    stdout when a communicator comes up, and a caller whose stdout is ONE JSON line (bench.py) must not carry it."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)


class Comm:
    """One RCCL communicator over the ranks of one node + the TCP side channel through rank 0."""

    def __init__(self, rank, world, device, addr=None, port=None, timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        self.hip = _lib('libamdhip64.so', 'APPLES_HIP_LIB')
        self.nccl = _lib('librccl.so', 'APPLES_RCCL_LIB')
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.nccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        self.nccl.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self.nccl.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self.nccl.ncclCommDestroy.argtypes = [C.c_void_p]
        self._check_hip(self.hip.hipSetDevice(int(device)), 'hipSetDevice')
        uid = _UniqueId()
        if self.rank == 0:
            self._check(self.nccl.ncclGetUniqueId(C.byref(uid)), 'ncclGetUniqueId')
        self.side = SideChannel(rank, world, bytes(uid.internal) if self.rank == 0 else None, addr, port, timeout)
        if self.rank != 0:
            if len(self.side.blob) != 128:
                raise RuntimeError('rendezvous delivered %d bytes in place of the 128-byte communicator id' % len(self.side.blob))
            C.memmove(C.byref(uid), self.side.blob, 128)
        self.comm = C.c_void_p()
        with _StdoutToStderr():
            self._check(self.nccl.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), 'ncclCommInitRank')
        self._recv_buf = C.c_void_p()
        self._recv_cap = 0
        self._host = C.c_void_p()
        self._host_cap = 0
        self.hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
        self.hip.hipHostFree.argtypes = [C.c_void_p]

    # ------------------------------------------------------------------ helpers
    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError('%s failed with ncclResult %d' % (what, rc))

    def _check_hip(self, rc, what):
        if rc != 0:
            raise RuntimeError('%s failed with hipError %d' % (what, rc))

    # ------------------------------------------------------------------ host-side barrier / reduction through rank 0
    def max_over_ranks(self, value):
        """max of one float over the ranks (every rank gets it); doubles as a barrier."""
        return self.side.max_over_ranks(value)

    def barrier(self):
        self.max_over_ranks(0.0)
        self._check_hip(self.hip.hipDeviceSynchronize(), 'hipDeviceSynchronize')

    # ------------------------------------------------------------------ the gather
    def gather_to_host(self, dev_ptr, sizes):
        """dev_ptr: this rank's device buffer of sizes[rank] bytes.  One grouped send/recv to rank 0, which
        returns the concatenation (in rank order) as a bytes-like view of the communicator's page-locked buffer (valid
        until the next gather; `bytes(view)` copies); the other ranks return None."""
        total = int(sum(sizes))
        if self.rank == 0 and total > self._recv_cap:
            if self._recv_buf:
                self.hip.hipFree(self._recv_buf)
            self._check_hip(self.hip.hipMalloc(C.byref(self._recv_buf), max(total, 1)), 'hipMalloc')
            self._recv_cap = total
        self._check(self.nccl.ncclGroupStart(), 'ncclGroupStart')
        if sizes[self.rank]:
            self._check(self.nccl.ncclSend(C.c_void_p(dev_ptr), sizes[self.rank], NCCL_UINT8, 0, self.comm, None), 'ncclSend')
        if self.rank == 0:
            off = 0
            for r in range(self.world):
                if sizes[r]:
                    self._check(self.nccl.ncclRecv(C.c_void_p(self._recv_buf.value + off), sizes[r], NCCL_UINT8, r, self.comm, None), 'ncclRecv')
                off += sizes[r]
        self._check(self.nccl.ncclGroupEnd(), 'ncclGroupEnd')
        self._check_hip(self.hip.hipDeviceSynchronize(), 'hipDeviceSynchronize')
        if self.rank != 0:
            return None
        # into a page-locked buffer that lives with the communicator (a pageable destination is staged through the driver's own
        # bounce buffer: 4 MB took 0.26 ms in place of 0.08); the caller gets a view of it, valid until the next gather
        if total > self._host_cap:
            if self._host:
                self.hip.hipHostFree(self._host)
            self._host = C.c_void_p()
            self._check_hip(self.hip.hipHostMalloc(C.byref(self._host), max(total, 1), 0), 'hipHostMalloc')
            self._host_cap = total
        self._check_hip(self.hip.hipMemcpy(self._host, self._recv_buf, total, 2), 'hipMemcpy')  # 2 = device to host
        return memoryview((C.c_char * total).from_address(self._host.value)).cast('B') if total else memoryview(b'')

    def close(self):
        if getattr(self, 'comm', None):
            self.nccl.ncclCommDestroy(self.comm)
            self.comm = None
        if self._recv_buf:
            self.hip.hipFree(self._recv_buf)
            self._recv_buf = C.c_void_p()
        if self._host:
            self.hip.hipHostFree(self._host)
            self._host = C.c_void_p()
            self._host_cap = 0
        self.side.close()
