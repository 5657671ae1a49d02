"""Worker for tests/test_distributed_gloo.py: one rank of a world_size-N gloo job.  Each rank
places its shard (CPU oracle stands in for the device, this is a test of the sharding + gather
path only) and rank 0 writes the gathered result."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def main():
    import torch.distributed as dist
    from apples_amd import synth
    from apples_amd.distributed import gather_placements, shard_bounds
    from apples_amd.engine import jc69_lut
    from oracle_c import COracle
    out_path, nq = sys.argv[1], int(sys.argv[2])
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    d = synth.make_dataset(400, 200, nq)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    co = COracle(d.tree, d.ref_seqs, nodes, method='OLS', lut=jc69_lut(200, 0.001))
    lo, hi = shard_bounds(nq, world)[rank]
    local = co.place_sequences(d.query_seqs[lo:hi])
    dist.barrier()
    full = gather_placements(local, nq, rank, world, dist)
    if rank == 0:
        np.save(out_path, full)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
