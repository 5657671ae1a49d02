#!/bin/bash
# The cost of the polytomy-capable (PL) instances of the lean sweep's kernels on a tree WITHOUT polytomies: config 3's binary
# backbone with the PL kernels forced (APPLES_LEAN_FORCE_POLY: 1 both, 2 bottom-up only, 3 top-down only), then the same with the
# kernels compiled for two wavefronts per SIMD.  Run on the GPU box: bash scripts/r06_poly_exp.sh > gpurun_out/r06_poly_exp.txt
cd $GRAFT_REPO_ROOT
leg() { python scripts/shape_legs.py c3 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin)['c3']; print('$1', round(d['ms_per_step'],2), d['per_kernel_ms_per_step'])"; }
leg plain
APPLES_LEAN_FORCE_POLY=1 leg forced_both
APPLES_LEAN_FORCE_POLY=2 leg forced_up
APPLES_LEAN_FORCE_POLY=3 leg forced_down
for V in "-DLEAN_UP_WAVES=2" "-DLEAN_UP_WAVES=2 -DLEAN_DOWN_WAVES=2" "-DLEAN_POLY_INLINE=__forceinline__"; do
  touch apples_amd/csrc/sweep_lean.hip
  APPLES_EXTRA_HIPCC_FLAGS="$V" python -m apples_amd.build > /dev/null 2>&1
  APPLES_LEAN_FORCE_POLY=1 leg "forced_both[$V]"
  leg "plain[$V]"
done
