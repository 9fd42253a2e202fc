#!/bin/bash
# small-grid kernel pair (pp_demons_small.h) against the marching kernels, by grid size
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r3
{
for sz in "43 43 22 100" "64 64 32 100" "85 85 43 100" "128 128 64 60" "171 171 86 40" "256 256 128 30"; do
  timeout 120 $KB $MAIN $sz "PP_FUSED_SMALL=0" "PP_FUSED_SMALL=100000000" "PP_FUSED_SMALL=0" "PP_FUSED_SMALL=100000000"
done
} 2>&1 | tee gpurun_out/r3/kbench_r3_9.txt
