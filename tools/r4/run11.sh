#!/bin/bash
# round 4, GPU visit 11: ESM selects on fresh compare masks (no scalar-written VCC in front of a v_cndmask)
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r4
{
for rep in 1 2 3; do
  for lib in $V/prev_main.so $MAIN; do
    timeout 60 $KB $lib 512 512 256 30 "PP_FUSED_MASK=1"
  done
done
timeout 60 $KB $MAIN 340 340 170 40 "PP_FUSED_MASK=1"
timeout 60 $KB $V/prev_main.so 340 340 170 40 "PP_FUSED_MASK=1"
timeout 60 $KB $MAIN 341 341 171 40 "PP_FUSED_MASK=1"
timeout 60 $KB $V/prev_main.so 341 341 171 40 "PP_FUSED_MASK=1"
} 2>&1 | tee gpurun_out/r4/kbench11.txt
timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "demons or fused" 2>&1 | tail -3
