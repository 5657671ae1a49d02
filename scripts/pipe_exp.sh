# Distance pass of sub-batch i+1 on part of the CUs beside selection + sweep of sub-batch i on the rest
# (APPLES_PIPELINE sub-batches, APPLES_GEMM_CUS persistent distance workgroups, APPLES_SWEEP_TEAMS sweep teams)
one() { timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu --timed resident 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2), d['roofline']['per_kernel_ms_per_step'])"; }
echo "== default: $(one)"
for cfg in "4 192 0" "8 192 0" "8 192 512" "8 176 640" "8 208 384" "16 192 512"; do
  set -- $cfg
  export APPLES_PIPELINE=$1 APPLES_GEMM_CUS=$2
  if [ "$3" != "0" ]; then export APPLES_SWEEP_TEAMS=$3; else unset APPLES_SWEEP_TEAMS; fi
  echo "== pipeline=$1 gemm_cus=$2 sweep_teams=$3: $(one)"
done
