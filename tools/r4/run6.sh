#!/bin/bash
# round 4, GPU visit 6: kernel A -- separate LDS objects x wavefront vote
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r4
{
for rep in 1 2 3; do
  for lib in $MAIN $V/nosplit.so $V/novoteA_split.so $V/novoteA_nosplit.so; do
    timeout 60 $KB $lib 512 512 256 30 "PP_FUSED_MASK=1"
  done
done
} 2>&1 | tee gpurun_out/r4/kbench6.txt
