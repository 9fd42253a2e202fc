#!/bin/bash
# streaming (nt) output stores vs cached ones across volume sizes
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
SO=platipy_amd/csrc/libplatipy_hip.so
{
for sz in "256 256 128" "340 341 171" "384 384 192" "448 448 224" "512 512 256"; do
timeout 120 $KB $SO $sz 20 "PP_FUSED_NT=0" "PP_FUSED_NT=1" "PP_FUSED_NT=0" "PP_FUSED_NT=1"
done
} 2>&1 | tee gpurun_out/kbench12.txt
