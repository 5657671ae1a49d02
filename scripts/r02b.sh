#!/bin/bash
# round-2 probes: gather cost model, HBM point of the distance kernel with its counter pass, CPU-leg check
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 $R/scripts/bin/gather_probe > $O/gather_probe.txt 2>&1
echo "gather rc=$?"
cat /sys/fs/cgroup/cpu.max > $O/cpu_max.txt 2>&1; nproc >> $O/cpu_max.txt
rocprofv3 -L > $O/counters.txt 2>&1
timeout 600 python3 $R/scripts/hbm_point_probe.py > $O/hbm_point.json 2> $O/hbm_point.err
echo "hbm rc=$?"; cat $O/hbm_point.json
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/hbm_fetch -- python3 $R/scripts/hbm_point_probe.py > $O/hbm_fetch.log 2>&1
echo "hbm pmc rc=$?"
timeout 600 python3 $R/bench.py --workload c2 --steps 5 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err
echo "bench c2 rc=$?"; python3 -c "import json;d=json.load(open('$O/bench_c2.json'));print(d['cpu_baseline'])"
