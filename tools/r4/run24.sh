#!/bin/bash
# round 4, GPU visit 24: mixed tile shapes (64 x 16 + one column of 32 x 32) against one shape, pipeline level sizes
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
export KB_SPACING=1.5,1.5,1.5
MAIN=platipy_amd/csrc/libplatipy_hip.so
timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "fused or demons" 2>&1 | tail -2
for size in "341 341 171 30" "340 340 170 30" "85 85 43 100" "405 405 200 20" "288 288 160 30" "171 171 85 60"; do
  echo "== $size"
  for rep in 1 2; do
    timeout 60 $KB $MAIN $size "PP_FUSED_MIX=1" "PP_FUSED_MIX=0"
  done
done 2>&1 | cut -c1-210
