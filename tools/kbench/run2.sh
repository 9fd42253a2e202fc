#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
KB=tools/kbench/kbench
{
for v in nl ng ns nlg nlgs ngs; do
timeout 120 $KB tools/kbench/variants/$v.so 512 512 256 20 "PP_FUSED_GEN=2"
done
} 2>&1 | tee gpurun_out/kbench2.txt
