"""Query sharding across the GPUs of one node and the single result gather.

Queries are independent (run_apples.py:101-102), so rank r places the contiguous block
[r*Q/W, (r+1)*Q/W) with no communication, and the 40-byte placement structs are gathered to rank 0
once -- over RCCL/xGMI with the "nccl" backend, over gloo in the CPU tests.  Input order is
preserved by construction, as starmap does.  torch.distributed is plumbing only."""
import numpy as np


def shard_bounds(n, world):
    """[(lo, hi)] per rank, contiguous, sizes differ by at most one."""
    return [(r * n // world, (r + 1) * n // world) for r in range(world)]


def gather_bytes(local, rank, world, dist, sizes=None):
    """Gather a 1-D uint8 torch tensor from every rank to rank 0.  `sizes` = per-rank lengths (all
    equal when None).  Returns the list of per-rank tensors on rank 0, None elsewhere."""
    import torch
    if sizes is None:
        sizes = [local.numel()] * world
    m = max(sizes)
    buf = local
    if local.numel() < m:
        buf = torch.zeros(m, dtype=torch.uint8, device=local.device)
        buf[:local.numel()] = local
    out = [torch.empty(m, dtype=torch.uint8, device=local.device) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    return [o[:s] for o, s in zip(out, sizes)]


def gather_placements(local, n_total, rank, world, dist, device='cpu'):
    """local: structured numpy array (engine.PLACEMENT_DTYPE) for this rank's shard of n_total
    queries.  Returns the full array in query order on rank 0, None elsewhere."""
    import torch
    bounds = shard_bounds(n_total, world)
    item = local.dtype.itemsize
    sizes = [(hi - lo) * item for lo, hi in bounds]
    assert len(local) * item == sizes[rank]
    t = torch.from_numpy(np.frombuffer(local.tobytes(), dtype=np.uint8).copy()).to(device)
    parts = gather_bytes(t, rank, world, dist, sizes)
    if parts is None:
        return None
    raw = b''.join(p.cpu().numpy().tobytes() for p in parts)
    return np.frombuffer(raw, dtype=local.dtype).copy()
