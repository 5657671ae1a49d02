#!/bin/bash
# three fuzz processes side by side on the one GPU (what `pytest -n 3` does), each with its own trace: which configuration and
# route a crash belongs to
#   bash scripts/r06_fuzz_parallel.sh 200 215 clustered
LO=$1; HI=$2; K=${3:-clustered}
N=$(( (HI - LO + 1 + 2) / 3 ))
for W in 0 1 2; do
  A=$(( LO + W * N )); B=$(( A + N - 1 )); [ $B -gt $HI ] && B=$HI
  [ $A -gt $HI ] && continue
  ( APPLES_FUZZ_TRACE=1 APPLES_FUZZ_SEEDS="$A-$B" timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x -s -k "$K" -p no:cacheprovider > gpurun_out/fuzz_w$W.log 2>&1; echo "worker $W ($A-$B) rc $?" ) &
done
wait
for W in 0 1 2; do echo "== worker $W"; grep -E "^cfg|^route|fault|passed|failed|Error|error" gpurun_out/fuzz_w$W.log | tail -6; done
