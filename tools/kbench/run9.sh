#!/bin/bash
# variants given on the command line against the product build, 512x512x256, one box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
KB=tools/kbench/kbench
{
timeout 120 $KB platipy_amd/csrc/libplatipy_hip.so 512 512 256 20 "PP_FUSED_SUM=1" "PP_FUSED_SUM=1"
for v in "$@"; do
timeout 120 $KB tools/kbench/variants/$v.so 512 512 256 20 "PP_FUSED_SUM=1"
done
timeout 120 $KB platipy_amd/csrc/libplatipy_hip.so 512 512 256 20 "PP_FUSED_SUM=1"
} 2>&1 | tee gpurun_out/kbench9.txt
