#!/bin/bash
cd "$(dirname "$0")/../.."
SB=tools/kbench/sbench
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r3
{
timeout 200 $SB $MAIN 512 512 256 5 | grep -E "compose"
echo "PP_WARP_LEGACY=1"; PP_WARP_LEGACY=1 timeout 200 $SB $MAIN 512 512 256 5 | grep -E "compose"
} 2>&1 | tee gpurun_out/r3/sbench_r3_7.txt
bash tools/gpu_reg_prof.sh 2>&1 | head -24 | cut -c1-170
