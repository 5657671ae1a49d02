for cfg in "64 8192" "64 5120" "64 4096" "64 3072" "64 2048"; do
  set -- $cfg
  export APPLES_SWEEP_TEAM=$1; export APPLES_SWEEP_TEAMS=${2}
  echo "== team=$1 teams=$2"; python bench.py --steps 5 --warmup 2 --no-cpu 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['per_kernel_ms_per_step'])"
done
