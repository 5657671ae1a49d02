# What the scoredist filter GEMM's time is made of (C4): the kernel with parts left out (wrong results: timing only).
# Builds variants of the library on the GPU box.  usage: bash scripts/r04_sd_parts_exp.sh
cd $GRAFT_REPO_ROOT
run() {
  APPLES_EXTRA_HIPCC_FLAGS="$1" python -m apples_amd.build --force > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/sdp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sdp -- python3 $GRAFT_REPO_ROOT/bench.py --workload c4 --no-cpu --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob('/tmp/sdp/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k_sd_gemm' in r['Name'] and 'false' in r['Name']:
        print('%-45s k_sd_gemm avg %8.3f ms' % ('$1' or 'as built', float(r['AverageNs']) / 1e6))
PY
  cd $GRAFT_REPO_ROOT
}
run ""
run "-DSD_SKIP_EPILOGUE"
run "-DSD_NO_DMA -DSD_SKIP_EPILOGUE"
run "-DSD_NO_FRAGS -DSD_SKIP_EPILOGUE"
run "-DSD_NO_DMA -DSD_NO_FRAGS -DSD_SKIP_EPILOGUE"
python -m apples_amd.build --force > /dev/null 2>&1
