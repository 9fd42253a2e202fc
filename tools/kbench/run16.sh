#!/bin/bash
# very short z-chunks on the coarse pyramid levels
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
SO=platipy_amd/csrc/libplatipy_hip.so
{
timeout 120 $KB $SO 85 85 43 40 "PP_FUSED_SUM=1" "PP_FUSED_ZCHUNK=1" "PP_FUSED_ZCHUNK=2" "PP_FUSED_ZCHUNK=3" "PP_FUSED_ZCHUNK=4" "PP_FUSED_SUM=1"
timeout 120 $KB $SO 171 171 86 40 "PP_FUSED_SUM=1" "PP_FUSED_ZCHUNK=2" "PP_FUSED_ZCHUNK=3" "PP_FUSED_ZCHUNK=4" "PP_FUSED_ZCHUNK=6" "PP_FUSED_ZCHUNK=8" "PP_FUSED_ZCHUNK=11" "PP_FUSED_SUM=1"
timeout 120 $KB $SO 43 43 22 40 "PP_FUSED_SUM=1" "PP_FUSED_ZCHUNK=1" "PP_FUSED_ZCHUNK=2"
} 2>&1 | tee gpurun_out/kbench16.txt
