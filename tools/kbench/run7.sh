#!/bin/bash
# tile shape and z-chunk sweep on the current build (SUM mode), one box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
KB=tools/kbench/kbench
SO=platipy_amd/csrc/libplatipy_hip.so
{
timeout 300 $KB $SO 512 512 256 20 "PP_FUSED_SUM=1" "PP_FUSED_TILE=1" "PP_FUSED_ZCHUNK=32" "PP_FUSED_ZCHUNK=43" "PP_FUSED_ZCHUNK=64" "PP_FUSED_ZCHUNK=86" "PP_FUSED_ZCHUNK=128" "PP_FUSED_TILE=1,PP_FUSED_ZCHUNK=128" "PP_FUSED_SUM=1"
} 2>&1 | tee gpurun_out/kbench7.txt
