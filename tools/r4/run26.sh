#!/bin/bash
# round 4, GPU visit 26: padded rows (16-byte aligned strips, MASK instances for odd rows) against dense rows
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
export KB_SPACING=1.5,1.5,1.5
MAIN=platipy_amd/csrc/libplatipy_hip.so
timeout 900 python -m pytest tests/test_kernels.py tests/test_registration.py -m gpu -x -q 2>&1 | tail -2
for size in "341 341 171 30" "342 342 171 30" "171 171 85 60" "85 85 43 100" "405 405 200 20"; do
  echo "== $size"
  for rep in 1 2; do
    timeout 60 $KB $MAIN $size "PP_FUSED_PITCH=1" "PP_FUSED_PITCH=0"
  done
done 2>&1 | cut -c1-215
timeout 300 python tools/r4/chain_levels.py
