#!/bin/bash
# variants at two sizes, interleaved, one box
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
{
for rep in 1 2; do
for v in platipy_amd/csrc/libplatipy_hip.so "$@"; do
  so=$v; [ -f "$so" ] || so=tools/kbench/variants/$v.so
  timeout 120 $KB $so 512 512 256 20 "PP_FUSED_SUM=1"
  timeout 120 $KB $so 341 341 171 20 "PP_FUSED_SUM=1"
done
done
} 2>&1 | tee gpurun_out/kbench14.txt
