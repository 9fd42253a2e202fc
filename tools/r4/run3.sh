#!/bin/bash
# round 4, GPU visit 3: the voted fast paths (kernel B: interior warp samples; kernel A: central-difference gradients).
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r4
{
for rep in 1 2 3; do
  for lib in $V/novote.so $V/novoteA.so $V/novoteB.so $MAIN; do
    timeout 120 $KB $lib 512 512 256 30 "PP_FUSED_MASK=1"
  done
done
timeout 120 $KB $MAIN 512 512 256 30 "PP_FUSED_MASK=0"
timeout 120 $KB $MAIN 340 340 170 40 "PP_FUSED_MASK=1"
timeout 120 $KB $V/novote.so 340 340 170 40 "PP_FUSED_MASK=1"
timeout 120 $KB $MAIN 341 341 171 40 "PP_FUSED_MASK=1"
timeout 120 $KB $V/novote.so 341 341 171 40 "PP_FUSED_MASK=1"
timeout 120 $KB $MAIN 128 128 64 40 "PP_FUSED_MASK=1"
timeout 120 $KB $V/novote.so 128 128 64 40 "PP_FUSED_MASK=1"
} 2>&1 | tee gpurun_out/r4/kbench3.txt
timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "demons or fused" 2>&1 | tail -8 | tee gpurun_out/r4/kernel_tests3.txt
