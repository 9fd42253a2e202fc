#!/bin/bash
# round 3, A/B 1: plain ESM path / scalar-offset addressing against the round-2 kernels (same box, same run)
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r3
{
for rep in 1 2; do
for so in $V/head_r2.so $MAIN $V/nofast.so $V/nosoff.so $V/nofast_nosoff.so; do
  timeout 120 $KB $so 512 512 256 30 "PP_FUSED_SUM=1"
done
timeout 120 $KB $MAIN 512 512 256 30 "PP_FUSED_PLAIN=0"
done
for so in $V/head_r2.so $MAIN; do
  timeout 120 $KB $so 341 341 171 30 "PP_FUSED_SUM=1"
  timeout 120 $KB $so 171 171 86 40 "PP_FUSED_SUM=1"
  timeout 120 $KB $so 85 85 43 40 "PP_FUSED_SUM=1"
done
} 2>&1 | tee gpurun_out/r3/kbench_r3_1.txt
