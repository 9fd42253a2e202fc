#!/bin/bash
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r3
{
for rep in 1 2; do
  timeout 120 $KB $MAIN 512 512 256 30 "PP_FUSED_A3=1" "PP_FUSED_A3=0"
  timeout 120 $KB $V/a3_noplain.so 512 512 256 30 "PP_FUSED_A3=1"
  timeout 120 $KB $V/a3_nosplit.so 512 512 256 30 "PP_FUSED_A3=1"
done
timeout 120 $KB $MAIN 341 341 171 30 "PP_FUSED_A3=1" "PP_FUSED_A3=0"
timeout 120 $KB $MAIN 85 85 43 40 "PP_FUSED_A3=1" "PP_FUSED_A3=0"
} 2>&1 | tee gpurun_out/r3/kbench_r3_4.txt
