#!/bin/bash
# 64 x 32 tiles / 1024-thread blocks per kernel
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
SO=platipy_amd/csrc/libplatipy_hip.so
{
for sz in "512 512 256" "341 341 171"; do
timeout 120 $KB $SO $sz 20 "PP_FUSED_SUM=1" "PP_FUSED_TILE_A=2" "PP_FUSED_TILE_B=2" "PP_FUSED_TILE_A=2,PP_FUSED_TILE_B=2" "PP_FUSED_SUM=1" "PP_FUSED_TILE_A=2"
done
} 2>&1 | tee gpurun_out/kbench15.txt
