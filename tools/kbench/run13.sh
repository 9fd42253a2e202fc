#!/bin/bash
# 32 x 16 tiles / 256-thread blocks (PP_FUSED_TILE=2) against the default shapes, several grids
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
SO=platipy_amd/csrc/libplatipy_hip.so
{
for sz in "512 512 256" "341 341 171" "171 171 86" "85 85 43" "256 256 128"; do
timeout 120 $KB $SO $sz 20 "PP_FUSED_SUM=1" "PP_FUSED_TILE=2" "PP_FUSED_SUM=1" "PP_FUSED_TILE=2"
done
} 2>&1 | tee gpurun_out/kbench13.txt
