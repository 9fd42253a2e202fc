#!/bin/bash
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r3
{
for rep in 1 2 3; do
  timeout 120 $KB $V/defer1.so 512 512 256 30 "PP_FUSED_SUM=1"
  timeout 120 $KB $MAIN 512 512 256 30 "PP_FUSED_SUM=1"
done
for sz in "341 341 171 30" "171 171 86 40" "85 85 43 40" "512 512 512 10"; do
  timeout 120 $KB $V/defer1.so $sz "PP_FUSED_SUM=1"
  timeout 120 $KB $MAIN $sz "PP_FUSED_SUM=1"
done
} 2>&1 | tee gpurun_out/r3/kbench_r3_8.txt
