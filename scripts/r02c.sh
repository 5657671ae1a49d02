#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_parity.log 2>&1; echo "parity rc=$?"; tail -15 $O/pytest_parity.log
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py > $O/pytest_rest.log 2>&1; echo "rest rc=$?"; tail -15 $O/pytest_rest.log
timeout 300 python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
python3 -c "import json;d=json.load(open('$O/bench_c2.json'));print('c2',d['value'],d['ms_per_step'],d['resident']['ms_per_step'],d['resident']['per_kernel_ms_per_step'])"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python3 -c "import json;d=json.load(open('$O/bench_c3.json'));print('c3',d['value'],d['ms_per_step'],d['resident']['ms_per_step'],d['resident']['per_kernel_ms_per_step'])"
tail -3 $O/bench_c3.err
