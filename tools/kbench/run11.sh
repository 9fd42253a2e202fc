#!/bin/bash
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
SO=platipy_amd/csrc/libplatipy_hip.so
{
for nx in 340 341 343 344; do
timeout 120 $KB $SO $nx 341 171 20 "PP_FUSED_SUM=1" "PP_FUSED_GEN=1"
done
timeout 120 $KB $SO 171 171 86 20 "PP_FUSED_SUM=1"
timeout 120 $KB $SO 85 85 43 20 "PP_FUSED_SUM=1"
} 2>&1 | tee gpurun_out/kbench11.txt
