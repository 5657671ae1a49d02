#!/bin/bash
# SQ / TCP counters of the block kernels on config 3's size clustered (one --pmc pass each; kernel trace only).
# usage on the GPU box: bash scripts/r05_blk_sq_pass.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_blk_sq
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "\b\(SQ\|TCP\|TA\|TCC\|GRBM\)_[A-Za-z0-9_]*" | sort -u > $OUT/counters.txt
run() { name=$1; shift; rm -rf /tmp/sqp; timeout 280 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/sqp -- python3 $R/bench.py --workload c3-clustered --no-cpu --no-extras --steps 1 --warmup 1 > $OUT/$name.log 2>&1
  python3 - "$name" <<'PY'
import csv, glob, sys, collections, re
f = glob.glob('/tmp/sqp/*/*counter_collection.csv')
if not f: print(sys.argv[1], 'no counter file'); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    m = re.search(r'k_[a-zA-Z0-9_]+(<[^>]*>)?', r['Kernel_Name']); k = m.group(0) if m else r['Kernel_Name'][:40]
    if not any(x in k for x in ('k_blocks', 'k_lean', 'k_cluster_dist', 'k_select_clusters<3')): continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k in acc:
    print(sys.argv[1], k, 'launches', len(cnt[k]), {c: round(v / len(cnt[k])) for c, v in acc[k].items()})
PY
}
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE | tee -a $OUT/summary.txt
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU | tee -a $OUT/summary.txt
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum | tee -a $OUT/summary.txt
