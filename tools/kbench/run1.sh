#!/bin/bash
# GPU batch 1: generation 1 vs 2, ring unroll vs moves, z-chunk / tile sweeps (writes gpurun_out/kbench1.txt)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
KB=tools/kbench/kbench
P=platipy_amd/csrc/libplatipy_hip.so
{
timeout 120 $KB $P 512 512 256 20 "PP_FUSED_GEN=1" "PP_FUSED_GEN=2" "PP_FUSED_GEN=1" "PP_FUSED_GEN=2"
timeout 120 $KB tools/kbench/variants/u0.so 512 512 256 20 "PP_FUSED_GEN=2"
timeout 120 $KB tools/kbench/variants/le.so 512 512 256 20 "PP_FUSED_GEN=2"
timeout 200 $KB $P 512 512 256 20 "PP_FUSED_ZCHUNK=16" "PP_FUSED_ZCHUNK=32" "PP_FUSED_ZCHUNK=43" "PP_FUSED_ZCHUNK=64" "PP_FUSED_ZCHUNK=128" "PP_FUSED_ZCHUNK=256" "PP_FUSED_TILE=1" "PP_FUSED_TILE=1,PP_FUSED_ZCHUNK=32"
timeout 120 $KB $P 341 341 171 20 "PP_FUSED_GEN=1" "PP_FUSED_GEN=2" "PP_FUSED_GEN=2,PP_FUSED_TILE=0"
timeout 120 $KB $P 512 512 512 10 "PP_FUSED_GEN=1" "PP_FUSED_GEN=2"
KB_SPACING=0.6,0.6,1.0 timeout 120 $KB $P 512 512 256 10 "PP_FUSED_GEN=1" "PP_FUSED_GEN=2"
} 2>&1 | tee gpurun_out/kbench1.txt
