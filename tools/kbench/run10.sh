#!/bin/bash
# row-length sensitivity around the pipelines' finest grid (341 x 341 x 171)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
KB=tools/kbench/kbench
SO=platipy_amd/csrc/libplatipy_hip.so
{
for nx in 320 336 340 341 344 352 384; do
timeout 120 $KB $SO $nx 341 171 20 "PP_FUSED_SUM=1"
done
timeout 120 $KB $SO 352 352 171 20 "PP_FUSED_SUM=1" "PP_FUSED_TILE=0"
timeout 120 $KB $SO 341 341 171 20 "PP_FUSED_SUM=1" "PP_FUSED_TILE=0" "PP_FUSED_TILE=1"
} 2>&1 | tee gpurun_out/kbench10.txt
