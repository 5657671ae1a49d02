#!/bin/bash
# k_cluster_dist (bit counts) and k_cluster_dist_mfma (matrix cores): the launch with and without its epilogue (table look-up + store per
# pair; without: results wrong, timing only).  usage on the GPU box: bash scripts/r05_cluster_dist_exp.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_cluster_dist_exp.txt
: > $OUT
cd $R
IFS="|" read -ra VS <<< "${VARIANTS:-|-DCL_EXP_NO_EPILOGUE}"
for v in "${VS[@]}"; do
  rm -f apples_amd/csrc/select.o
  APPLES_EXTRA_HIPCC_FLAGS="$v" python -m apples_amd.build > /dev/null 2>&1
  for e in "" "APPLES_NO_CLUSTER_MFMA=1"; do
    cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/cm
    env $e rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cm -- python3 $R/bench.py --workload c3-clustered --no-cpu --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
    echo "[$v] [$e] $(grep -h "k_cluster_dist" /tmp/cm/*/*kernel_stats.csv | cut -d, -f1,2,4)" | tee -a $OUT
    cd $R
  done
done
rm -f apples_amd/csrc/select.o; python -m apples_amd.build > /dev/null 2>&1
