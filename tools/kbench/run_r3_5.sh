#!/bin/bash
# round 3: stand-alone kernels (baseline of the round) + recursive Gaussian vs number of resident blocks
cd "$(dirname "$0")/../.."
SB=tools/kbench/sbench
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r3
{
timeout 200 $SB $MAIN 512 512 256 5
for g in 4096 2048 1024 512 256; do
  echo "PP_RG_GRID=$g"; PP_RG_GRID=$g timeout 200 $SB $MAIN 512 512 256 5 | grep recursive
done
} 2>&1 | tee gpurun_out/r3/sbench_r3_5.txt
