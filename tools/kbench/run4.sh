#!/bin/bash
# size / spacing sweep of the current build: generation 1 vs 2
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
KB=tools/kbench/kbench
P=platipy_amd/csrc/libplatipy_hip.so
{
timeout 120 $KB $P 512 512 256 20 "PP_FUSED_GEN=1" "PP_FUSED_GEN=2"
timeout 120 $KB $P 341 341 171 20 "PP_FUSED_GEN=1" "PP_FUSED_GEN=2" "PP_FUSED_GEN=2,PP_FUSED_TILE=0" "PP_FUSED_GEN=2,PP_FUSED_TILE=1"
timeout 120 $KB $P 512 512 512 10 "PP_FUSED_GEN=1" "PP_FUSED_GEN=2"
timeout 120 $KB $P 128 128 64 20 "PP_FUSED_GEN=1" "PP_FUSED_GEN=2"
timeout 120 $KB $P 64 64 32 20 "PP_FUSED_GEN=1" "PP_FUSED_GEN=2"
KB_SPACING=0.9766,0.9766,2.5 timeout 120 $KB $P 512 512 128 10 "PP_FUSED_GEN=1" "PP_FUSED_GEN=2"
KB_SPACING=0.7,0.7,1.0 timeout 120 $KB $P 512 512 256 10 "PP_FUSED_GEN=1" "PP_FUSED_GEN=2"
KB_SPACING=0.6,0.6,1.0 timeout 120 $KB $P 512 512 256 10 "PP_FUSED_GEN=1" "PP_FUSED_GEN=2"
KB_SPACING=0.5,0.5,0.5 timeout 120 $KB $P 512 512 256 10 "PP_FUSED_GEN=1" "PP_FUSED_GEN=2"
} 2>&1 | tee gpurun_out/kbench4.txt
