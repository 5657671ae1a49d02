"""The N>1 path (contiguous query shards + one gather to rank 0) on CPU: world_size 2 and 3 over
gloo, compared with the single-process result."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT

sys.path.insert(0, os.path.join(ROOT, 'oracle'))


@pytest.mark.parametrize('world,nq', [(2, 37), (3, 20)])
def test_sharded_gather_equals_single_process(tmp_path, world, nq):
    from apples_amd import synth
    from apples_amd.engine import jc69_lut
    from oracle_c import COracle
    out = tmp_path / 'gathered.npy'
    port = 29500 + (os.getpid() % 2000) + world
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'tests', '_dist_worker.py'),
           str(out), str(nq)]
    env = dict(os.environ, OMP_NUM_THREADS='1')
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = np.load(out)
    d = synth.make_dataset(400, 200, nq)
    nodes = np.array([d.tree.name_to_node[n] for n in d.ref_names], np.int32)
    want = COracle(d.tree, d.ref_seqs, nodes, method='OLS', lut=jc69_lut(200, 0.001)).place_sequences(d.query_seqs)
    assert got.tobytes() == want.tobytes()  # order preserved, nothing lost at ragged shard edges
