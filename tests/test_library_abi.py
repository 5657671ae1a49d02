"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/apples_hip.h declares; struct layouts agree between C and the ctypes mirror."""
import ctypes
import os
import re

import numpy as np

from helpers import ROOT


def _declared_symbols(header='apples_hip.h'):
    text = open(os.path.join(ROOT, 'include', header)).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(apples_[a-z_]+)\s*\(', text)))


def test_header_declares_the_documented_entry_points():
    syms = _declared_symbols()
    for s in ('apples_ctx_create', 'apples_ctx_destroy', 'apples_distances', 'apples_place_from_sequences',
              'apples_place_from_distances', 'apples_last_error'):
        assert s in syms


def test_library_builds_loads_and_exports_every_symbol():
    from apples_amd import build, engine
    lib_path = build.build(verbose=False)
    assert os.path.exists(lib_path)
    lib = ctypes.CDLL(lib_path)
    for s in _declared_symbols():
        assert hasattr(lib, s), 'missing export %s' % s
    assert sorted(engine.EXPORTS) == _declared_symbols()
    engine.load_library()


def test_io_library_builds_and_exports_every_symbol():
    """include/apples_io.h: the host-side scanner library (plain g++, no device code)."""
    from apples_amd import build
    lib = ctypes.CDLL(build.build_io(verbose=False))
    syms = _declared_symbols('apples_io.h')
    assert syms == ['apples_consensus', 'apples_dismat_scan', 'apples_extended_newick', 'apples_fasta_scan', 'apples_fasta_scan_mt', 'apples_format_double',
                    'apples_jplace_rows', 'apples_max_clusters', 'apples_newick_scan']
    for s in syms:
        assert hasattr(lib, s), 'missing export %s' % s


def test_every_header_in_include_is_covered():
    assert sorted(os.listdir(os.path.join(ROOT, 'include'))) == ['apples_hip.h', 'apples_io.h']


def test_placement_struct_layout_matches_header():
    from apples_amd.engine import PLACEMENT_DTYPE
    assert PLACEMENT_DTYPE.itemsize == 40
    assert [PLACEMENT_DTYPE.fields[n][1] for n in PLACEMENT_DTYPE.names] == [0, 4, 8, 16, 24, 32, 36]


def test_missing_gpu_fails_loudly():
    """No silent CPU path: without a device the context constructor raises."""
    import subprocess
    import sys
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from apples_amd.tree import parse_newick\n"
            "from apples_amd.engine import Engine\n"
            "t = parse_newick('((A:0.1,B:0.2):0.25,(C:0.3,(D:0.2,E:0.2):0.2):0.25);')\n"
            "try:\n"
            "    Engine(t, None)\n"
            "except RuntimeError as e:\n"
            "    print('RAISED', e)\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES='-1', ROCR_VISIBLE_DEVICES='')
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=300)
    assert 'RAISED' in out.stdout, out.stdout + out.stderr


def test_jc69_table_matches_reference_expression():
    """The host-filled JC69 table equals the oracle's scalar evaluation for every integer pair."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import apples_oracle as orc
    from apples_amd.engine import jc69_lut
    for L, V in ((37, 0.001), (64, 0.5)):
        lut = jc69_lut(L, V)
        assert len(lut) == (L + 1) * (L + 2) // 2
        for valid in range(L + 1):
            for mism in range(valid + 1):
                want = orc.jc69_from_counts(mism, valid, L, V)
                got = lut[valid * (valid + 1) // 2 + mism]
                assert got == want and np.signbit(got) == np.signbit(want), (mism, valid)


def test_params_struct_layout_matches_header(tmp_path):
    """apples_params as the C compiler lays it out (gcc on include/apples_hip.h) against the ctypes mirror,
    the debug switches included."""
    import subprocess
    from apples_amd import engine
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "apples_hip.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(apples_params), offsetof(apples_params, filt_threshold), '
                   'offsetof(apples_params, jc_lut), offsetof(apples_params, max_batch), offsetof(apples_params, debug), offsetof(apples_params, batch_gib)); '
                   'printf("%u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u\\n", APPLES_DBG_NO_FUSE, APPLES_DBG_SWEEP_SCAN, APPLES_DBG_NODE_MAP, APPLES_DBG_SWEEP_MERGE, '
                   'APPLES_DBG_NO_SWEEP_MERGE, APPLES_DBG_NO_DIST_GEMM, APPLES_DBG_NO_SWEEP_LEAN, APPLES_DBG_NO_SD_GEMM, APPLES_DBG_CLUSTER_BY_QUERY, '
                   'APPLES_DBG_NO_CLUSTER_TOPUP, APPLES_DBG_NO_STREAM_SELECT, APPLES_DBG_NO_TOPUP_KERNEL, APPLES_DBG_NO_CLUSTER_BIG, '
                   'APPLES_DBG_NO_SD_TOPUP, APPLES_DBG_SD_FP6, APPLES_DBG_NO_TOPUP_OVERLAP, APPLES_DBG_STREAM_THIRD_PASS, APPLES_DBG_NO_SD_COMPACT, APPLES_DBG_SD_COMPACT_TINY, APPLES_DBG_NO_BLOCKS, APPLES_DBG_HYBRID_RECORDS, APPLES_DBG_NO_CLUSTER_MFMA, APPLES_DBG_ALL); return 0; }\n')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split('\n')
    size, o_thr, o_lut, o_mb, o_dbg, o_gib = map(int, out[0].split())
    P = engine._Params
    assert ctypes.sizeof(P) == size
    assert (P.filt_threshold.offset, P.jc_lut.offset, P.max_batch.offset, P.debug.offset, P.batch_gib.offset) == (o_thr, o_lut, o_mb, o_dbg, o_gib)
    bits = list(map(int, out[1].split()))
    names = ('no_fuse', 'sweep_scan', 'node_map', 'sweep_merge', 'no_sweep_merge', 'no_dist_gemm', 'no_sweep_lean', 'no_sd_gemm',
             'cluster_by_query', 'no_cluster_topup', 'no_stream_select', 'no_topup_kernel', 'no_cluster_big', 'no_sd_topup', 'sd_fp6', 'no_topup_overlap', 'stream_third_pass', 'no_sd_compact', 'sd_compact_tiny', 'no_blocks', 'hybrid_records', 'no_cluster_mfma')
    assert bits[:-1] == [engine.DBG[k] for k in names]
    assert bits[-1] == sum(engine.DBG.values()) and len(engine.DBG) == len(names)  # APPLES_DBG_ALL = every switch
