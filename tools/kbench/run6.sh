#!/bin/bash
# SUM mode (kernel A stores D + U) against separate volumes, same box, same run.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
KB=tools/kbench/kbench
SO=platipy_amd/csrc/libplatipy_hip.so
{
timeout 120 $KB $SO 512 512 256 20 "PP_FUSED_SUM=0" "PP_FUSED_SUM=1" "PP_FUSED_SUM=0" "PP_FUSED_SUM=1"
timeout 120 $KB $SO 341 341 171 20 "PP_FUSED_SUM=0" "PP_FUSED_SUM=1"
timeout 120 $KB $SO 512 512 512 10 "PP_FUSED_SUM=0" "PP_FUSED_SUM=1"
for v in "$@"; do
timeout 120 $KB tools/kbench/variants/$v.so 512 512 256 20 "PP_FUSED_SUM=1" "PP_FUSED_SUM=1"
timeout 120 $KB tools/kbench/variants/$v.so 341 341 171 20 "PP_FUSED_SUM=1"
done
} 2>&1 | tee gpurun_out/kbench6.txt
