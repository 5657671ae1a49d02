for cfg in "0 16" "1 16" "0 32" "0 8"; do
  set -- $cfg
  unset APPLES_BCNT_ASM
  [ "$1" = "1" ] && export APPLES_BCNT_ASM=1
  export APPLES_DIST_TILE=$2
  echo "== asm=$1 tile=$2"
  python bench.py --steps 5 --warmup 2 --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['per_kernel_ms_per_step'])"
  python scripts/c3_probe.py 8192 2>&1 | tail -3 | head -1
done
