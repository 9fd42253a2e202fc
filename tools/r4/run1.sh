#!/bin/bash
# round 4, GPU visit 1: do the MASK kernels (branch-free memory instructions) move the demons iteration, and is their
# out-of-range store mask honoured by the hardware?  kbench variants + the demons kernel tests + the instruction-rate probe.
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r4
{
echo "=== kbench 512x512x256 (cs = field checksum: equal = bit-identical)"
for rep in 1 2; do
  timeout 120 $KB $MAIN 512 512 256 30 "PP_FUSED_MASK=0" "PP_FUSED_MASK=1"
  for lib in $V/finbefore.so $V/steadyB.so $V/nodefer.so; do
    timeout 120 $KB $lib 512 512 256 30 "PP_FUSED_MASK=1"
  done
done
echo "=== kbench 340x340x170 / 341x341x171 (pipeline level; odd rows keep the branchy kernels)"
timeout 120 $KB $MAIN 340 340 170 40 "PP_FUSED_MASK=0" "PP_FUSED_MASK=1"
timeout 120 $KB $V/steadyB.so 340 340 170 40 "PP_FUSED_MASK=1"
timeout 120 $KB $MAIN 341 341 171 40 "PP_FUSED_MASK=1"
echo "=== kbench 128x128x64, 64x64x32"
timeout 120 $KB $MAIN 128 128 64 40 "PP_FUSED_MASK=0" "PP_FUSED_MASK=1"
timeout 120 $KB $MAIN 64 64 32 40 "PP_FUSED_MASK=0" "PP_FUSED_MASK=1"
} 2>&1 | tee gpurun_out/r4/kbench1.txt
timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "demons or fused" 2>&1 | tail -8 | tee gpurun_out/r4/kernel_tests1.txt
timeout 200 tools/probes/valu_rate 2>&1 | tee gpurun_out/r4/valu_rate.txt
