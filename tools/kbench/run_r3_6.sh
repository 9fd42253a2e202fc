#!/bin/bash
cd "$(dirname "$0")/../.."
SB=tools/kbench/sbench
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r3
{
timeout 200 $SB $MAIN 512 512 256 5 | grep -E "recursive|compose|identity"
echo "PP_RG_TWO_SWEEP=1"; PP_RG_TWO_SWEEP=1 timeout 200 $SB $MAIN 512 512 256 5 | grep recursive
timeout 200 $SB $MAIN 341 341 171 5 | grep -E "recursive"
PP_RG_TWO_SWEEP=1 timeout 200 $SB $MAIN 341 341 171 5 | grep recursive
} 2>&1 | tee gpurun_out/r3/sbench_r3_6.txt
