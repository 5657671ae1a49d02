#!/bin/bash
# k_blocks_up by leaving parts out (results are wrong in the variants: timing only); usage on the GPU box: bash scripts/r05_blk_parts_exp.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_blk_parts
mkdir -p $OUT
cd $R
for v in "" "-DBLK_EXP_NO_STORE" "-DBLK_EXP_NO_LOAD" "-DBLK_EXP_NO_STORE -DBLK_EXP_NO_LOAD"; do
  rm -f apples_amd/csrc/sweep_lean.o
  APPLES_EXTRA_HIPCC_FLAGS="$v" python -m apples_amd.build > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp
  rm -rf $OUT/stats
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/scripts/r05_c3cl_probe.py > /dev/null 2> $OUT/log.txt
  echo "== [$v]: $(grep -h "k_blocks_up\|k_blocks_down" $OUT/stats/*/*kernel_stats.csv | cut -d, -f1,4 | sed 's/(anonymous namespace):://' | tr '\n' ' ')" | tee -a $OUT/summary.txt
  cd $R
done
rm -f apples_amd/csrc/sweep_lean.o; python -m apples_amd.build > /dev/null 2>&1
