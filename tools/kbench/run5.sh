#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
KB=tools/kbench/kbench
{
timeout 120 $KB platipy_amd/csrc/libplatipy_hip.so 512 512 256 20 "PP_FUSED_GEN=2" "PP_FUSED_GEN=2"
for v in "$@"; do
timeout 120 $KB tools/kbench/variants/$v.so 512 512 256 20 "PP_FUSED_GEN=2" "PP_FUSED_GEN=2"
timeout 120 $KB tools/kbench/variants/$v.so 341 341 171 20 "PP_FUSED_GEN=2"
done
timeout 120 $KB platipy_amd/csrc/libplatipy_hip.so 341 341 171 20 "PP_FUSED_GEN=2"
} 2>&1 | tee gpurun_out/kbench5.txt
