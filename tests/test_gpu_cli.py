"""End-to-end drop-in check: this build's run_apples.py against jplace files written by the
REFERENCE's run_apples.py (tests/golden/g7_cli_*.jplace).  Needs an MI355X."""
import json
import os
import subprocess
import sys

import pytest

from helpers import DATA, GOLD, ROOT, assert_prow

pytestmark = pytest.mark.gpu

# The reference runs behind these fixtures consumed this repo's own cluster table through a
# TreeCluster.py stand-in (tests/golden/make_goldens.py:g7), so cluster handling -- consensus
# representatives, heap-ordered expansion -- is compared end to end.
RUNS = {
    'aln_OLS': ['-s', 'ref.fa', '-q', 'query.fa', '-t', 'backbone.nwk', '-m', 'OLS', '-D', '-T', '2'],
    'aln_default': ['-s', 'ref.fa', '-q', 'query.fa', '-t', 'backbone.nwk', '-D', '-T', '2'],
    'aln_f03_b5_BME': ['-s', 'ref.fa', '-q', 'query.fa', '-t', 'backbone.nwk', '-m', 'BME', '-f', '0.3', '-b', '5', '-D',
                       '-T', '2'],
    'aln_OLS_singletons': ['-s', 'ref.fa', '-q', 'query.fa', '-t', 'backbone.nwk', '-m', 'OLS', '-D', '-T', '2',
                           '--no-clusters'],
    'dist_default': ['-d', 'dist.mat', '-t', 'backbone.nwk', '-T', '2'],
    'small_BME': ['-d', 'small_dist.mat', '-t', 'small_backbone.nwk', '-m', 'BME', '-T', '1'],
    # -p on its default route (clusters of this repo's table, consensus of the 21-symbol alphabet); inputs: tests/prot_cases.py
    'prot_default': ['-p', '-s', 'prot_ref.fa', '-q', 'prot_query.fa', '-t', 'prot_backbone.nwk', '-D', '-T', '2'],
    'prot_OLS_f01_b5': ['-p', '-s', 'prot_ref.fa', '-q', 'prot_query.fa', '-t', 'prot_backbone.nwk', '-m', 'OLS', '-f', '0.1',
                        '-b', '5', '-D', '-T', '2'],
}


@pytest.mark.parametrize('label', sorted(RUNS))
def test_cli_matches_reference_jplace(label, tmp_path):
    if label.startswith('prot_'):
        import prot_cases
        made = {os.path.basename(p): p for p in prot_cases.write_case(str(tmp_path))}
        args = [made.get(a, a) for a in RUNS[label]]
    else:
        args = [a if a.startswith('-') or not os.path.exists(os.path.join(DATA, a)) else os.path.join(DATA, a)
                for a in RUNS[label]]
    out = tmp_path / 'out.jplace'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'run_apples.py')] + args + ['-o', str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = json.load(open(out))
    want = json.load(open(os.path.join(GOLD, 'g7_cli_%s.jplace' % label)))
    assert list(got) == list(want) == ['fields', 'metadata', 'placements', 'tree', 'version']
    assert got['fields'] == want['fields'] and got['version'] == want['version'] and got['tree'] == want['tree']
    assert [p['n'] for p in got['placements']] == [p['n'] for p in want['placements']]
    for g, w in zip(got['placements'], want['placements']):
        assert_prow(g['p'][0], w['p'][0], ctx='%s %s' % (label, w['n'][0]))
    # text-level layout: sort_keys + indent=4 + trailing newline (run_apples.py:116-117)
    text = open(out).read()
    assert text.endswith('}\n') and text.startswith('{\n    "fields": [')


def test_cli_stdout_and_extended_reference(tmp_path):
    ext = tmp_path / 'ext.fa'
    with open(ext, 'w') as f:
        f.write(open(os.path.join(DATA, 'ref.fa')).read())
        f.write(open(os.path.join(DATA, 'query.fa')).read())
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'run_apples.py'), '-s', os.path.join(DATA, 'ref.fa'), '-x',
                        str(ext), '-t', os.path.join(DATA, 'backbone.nwk'), '-m', 'OLS', '-D'],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    want = json.load(open(os.path.join(GOLD, 'g7_cli_aln_OLS.jplace')))
    assert [p['n'] for p in got['placements']] == [p['n'] for p in want['placements']]
    for g, w in zip(got['placements'], want['placements']):
        assert_prow(g['p'][0], w['p'][0])


def test_cli_database_cache_matches_reference_jplace(tmp_path):
    """build_applesdtb.py + run_apples.py -a (the reference's database route, build_applesdtb.py:23-28,
    run_apples.py:25-35,69-75) against the reference's own jplace for the same inputs."""
    db = tmp_path / 'ref.dtb'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'build_applesdtb.py'), '-s', os.path.join(DATA, 'ref.fa'), '-t',
                        os.path.join(DATA, 'backbone.nwk'), '-o', str(db), '-D'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    out = tmp_path / 'out.jplace'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'run_apples.py'), '-a', str(db), '-q', os.path.join(DATA, 'query.fa'),
                        '-m', 'OLS', '-o', str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = json.load(open(out))
    want = json.load(open(os.path.join(GOLD, 'g7_cli_aln_OLS.jplace')))
    assert got['tree'] == want['tree']
    assert [p['n'] for p in got['placements']] == [p['n'] for p in want['placements']]
    for g, w in zip(got['placements'], want['placements']):
        assert_prow(g['p'][0], w['p'][0], ctx='database %s' % w['n'][0])


def test_bench_gather_through_librccl_without_torch():
    """bench.py --gather rccl: the multi-rank code path with the end-of-run gather as one grouped
    ncclSend/ncclRecv through ctypes (apples_amd/rccl.py), no PyTorch in the process; one rank, one GPU."""
    env = dict(os.environ, APPLES_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0', RANK='0', LOCAL_RANK='0',
               WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29541')
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '1', '--steps', '2', '--warmup', '1', '--workload', 'small',"
            " '--no-cpu', '--gather', 'rccl']\n"
            "runpy.run_path(%r, run_name='__main__')\n"
            "assert 'torch' not in sys.modules, 'PyTorch was imported'\n" % os.path.join(ROOT, 'bench.py'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and 'librccl' in line['config']['gather']
    assert r.stdout.strip().splitlines()[-1].startswith('{')  # the JSON line is the last thing on stdout


@pytest.mark.parametrize('scaling', ['weak', 'strong'])
def test_bench_multi_rank_path_on_one_gpu(scaling):
    """bench.py's multi-rank path (process group over RCCL, zero-copy view of the device-resident
    placements, gather to rank 0, max over ranks), launched the way the driver launches it, with a
    single rank so that one GPU is enough: the contract line must come out, and bench.py itself
    checks that what the gather delivers is what the device buffer holds."""
    env = dict(os.environ, APPLES_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', '29533' if scaling == 'weak' else '29534', os.path.join(ROOT, 'bench.py'),
           '--gpus', '1', '--steps', '2', '--warmup', '1', '--workload', 'small', '--no-cpu', '--scaling', scaling]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['steps'] == 2 and line['scaling'] == scaling
    assert line['value'] > 0 and line['roofline']['frac'] > 0 and line['resident']['value'] >= line['value'] * 0.5
    assert line['config']['queries_total'] == 2048
    assert line['config']['placed'] > 0


def _device_count():
    import ctypes
    try:
        hip = ctypes.CDLL('libamdhip64.so')
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def test_bench_two_ranks_when_two_gpus():
    """`bench.py --gpus 2` the way a user starts it (apples_amd/launcher.py: one process per GPU, the grouped ncclSend / ncclRecv
    of apples_amd/rccl.py between two real peers) against the same job on one rank: the same placement bytes on rank 0.
    Arms itself on the first box with two devices; this pool's boxes have one (skipped there)."""
    if _device_count() < 2:
        pytest.skip('needs two visible GPUs')
    crc = {}
    for n in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '2', '--warmup', '1', '--workload',
                            'small', '--no-cpu', '--no-extras', '--scaling', 'strong', '--gather', 'rccl'],
                           capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line['n_gpus'] == n and line['config']['queries_total'] == 2048
        crc[n] = line['config']['placements_crc32']
    assert crc[1] == crc[2]


def test_cli_reference_alignment_with_rows_beyond_the_tree(tmp_path):
    """-s holds three rows that are not backbone leaves, -x the extended alignment: the reference
    (fixture g7_cli_aln_superset.jplace, written by its own run_apples.py) never compares those rows
    with a query but keeps them out of the query set.  Default route (clusters built from the tree)."""
    sup = tmp_path / 'superset_ref.fa'
    with open(sup, 'w') as f:  # as tests/golden/make_goldens.py:superset_alignment
        f.write(open(os.path.join(DATA, 'ref.fa')).read())
        f.write(''.join('>' + r for r in open(os.path.join(DATA, 'query.fa')).read().split('>')[1:4]))
    ext = tmp_path / 'extended_ref.fa'
    with open(ext, 'w') as f:
        f.write(open(os.path.join(DATA, 'ref.fa')).read() + open(os.path.join(DATA, 'query.fa')).read())
    want = json.load(open(os.path.join(GOLD, 'g7_cli_aln_superset.jplace')))
    for extra in ([], ['--no-clusters']):
        out = tmp_path / ('out%d.jplace' % len(extra))
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'run_apples.py'), '-s', str(sup), '-x', str(ext), '-t',
                            os.path.join(DATA, 'backbone.nwk'), '-m', 'OLS', '-D', '-o', str(out)] + extra,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        got = json.load(open(out))
        assert [p['n'] for p in got['placements']] == [p['n'] for p in want['placements']]
        if not extra:  # the fixture's clusters are this build's default clusters
            for g, w in zip(got['placements'], want['placements']):
                assert_prow(g['p'][0], w['p'][0], ctx='superset %s' % w['n'][0])


def test_cli_rejects_a_query_alignment_of_another_length(tmp_path):
    q = tmp_path / 'short.fa'
    with open(q, 'w') as f:
        f.write('>q1\nACGTACGT\n>q2\nACGTACGA\n')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'run_apples.py'), '-s', os.path.join(DATA, 'ref.fa'), '-q', str(q),
                        '-t', os.path.join(DATA, 'backbone.nwk'), '-D'], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and 'sites' in r.stderr


def test_worker_with_two_device_entries_matches_one(tmp_path):
    """worker._run_sharded: one host thread and one engine context per device entry, contiguous query
    shards, host concatenation.  Two entries naming device 0 exercise the threaded path on a one-GPU
    box (every C-ABI entry sets its context's device itself); set_options from the main thread
    afterwards must land on the contexts' device."""
    import numpy as np
    from apples_amd import synth
    from apples_amd.fasta import Alignment
    from apples_amd.options import options_config
    from apples_amd.reference import ReducedReference
    from apples_amd.worker import QueryWorker
    d = synth.make_dataset(1500, 300, 301)
    opts, _ = options_config(['-t', 'x', '-s', 'r', '-q', 'q', '-m', 'OLS'])
    ref = ReducedReference(Alignment(d.ref_names, d.ref_seqs), False, None)
    one = QueryWorker(d.tree, opts, ref, devices=(0,))
    n1, r1 = one.run_sequences(d.query_names, d.query_seqs, rows=True)
    one.close()
    two = QueryWorker(d.tree, opts, ref, devices=(0, 0, 0))
    n2, r2 = two.run_sequences(d.query_names, d.query_seqs, rows=True)
    assert n1 == n2 and r1 == r2 and len(two._engines) == 3
    for e in two._engines.values():
        e.set_options(method='FM')
    opts.method_name = 'FM'
    n3, r3 = two.run_sequences(d.query_names, d.query_seqs, rows=True)
    two.close()
    fm = QueryWorker(d.tree, opts, ref, devices=(0,))
    n4, r4 = fm.run_sequences(d.query_names, d.query_seqs, rows=True)
    fm.close()
    assert r3 == r4 and r3 != r1


def test_cli_reestimates_the_backbone(tmp_path):
    """Without -D the reference re-estimates the backbone's branch lengths with FastTree before placing
    (apples/prepareTree.py:20-21).  Here: a stand-in executable that doubles every length -- the placements
    must be made on the tree it returns; with no executable the lengths are estimated on the GPU and equal the
    bundled FastTree's output on the same inputs (tests/golden/g9_fasttree_data.nwk) to its printed digits."""
    import stat
    stub = tmp_path / 'FastTree'
    with open(stub, 'w') as f:
        f.write('#!%s\nimport sys\nsys.path.insert(0, %r)\nfrom apples_amd import reestimate as R\n'
                'a = sys.argv\nt = R.from_newick(open(a[a.index("-intree") + 1]).read())\nstack = [t]\n'
                'while stack:\n    v = stack.pop(); stack.extend(v.children)\n    v.length = None if v.length is None else 2 * v.length\n'
                'print(R.to_newick(t))\n' % (sys.executable, ROOT))
    os.chmod(stub, os.stat(stub).st_mode | stat.S_IEXEC)
    base = [sys.executable, os.path.join(ROOT, 'run_apples.py'), '-s', os.path.join(DATA, 'ref.fa'), '-q',
            os.path.join(DATA, 'query.fa'), '-t', os.path.join(DATA, 'backbone.nwk'), '-m', 'OLS', '--no-clusters']
    env = dict(os.environ, PATH=os.path.dirname(sys.executable) + ':/usr/bin:/bin')
    env.pop('APPLES_FASTTREE', None)
    outs = {}
    for label, extra, e in (('D', ['-D'], env), ('none', [], env), ('stub', ['--fasttree', str(stub)], env)):
        out = tmp_path / (label + '.jplace')
        r = subprocess.run(base + extra + ['-o', str(out)], capture_output=True, text=True, timeout=600, env=e)
        assert r.returncode == 0, r.stderr
        outs[label] = (json.load(open(out)), r.stderr)
    import re
    from apples_amd import reestimate as R
    from test_fasttree_me import GOLD, splits
    assert outs['none'][0]['tree'] != outs['D'][0]['tree'] and len(outs['none'][0]['placements']) == 10
    have = splits(R.from_newick(re.sub(r'\{\d+\}', '', outs['none'][0]['tree'])))
    want = splits(R.from_newick(open(os.path.join(GOLD, 'g9_fasttree_data.nwk')).read()))
    assert set(have) == set(want) and max(abs(have[k] - want[k]) for k in want) <= 1.01e-5
    assert outs['stub'][0]['tree'] != outs['D'][0]['tree'] and len(outs['stub'][0]['placements']) == 10
    # every branch twice as long and the observed distances unchanged: other optima, same format
    for p in outs['stub'][0]['placements']:
        assert len(p['p'][0]) == 5 and p['p'][0][0] >= 0
