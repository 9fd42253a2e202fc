#!/bin/bash
# round 4, GPU visit 5: ablations of both kernels on the current build (what does each memory-instruction class cost now?) + barrier trace
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r4
{
for rep in 1 2; do
  for lib in $MAIN $V/abl_nogather.so $V/abl_nostore.so $V/abl_noload.so $V/ablB_nomem.so $V/ablA_noload.so $V/ablA_nostore.so $V/ablA_nomem.so; do
    timeout 60 $KB $lib 512 512 256 30 "PP_FUSED_MASK=1"
  done
done
timeout 60 $KB $V/trace.so 512 512 256 12 "PP_FUSED_MASK=1"
timeout 60 $KB $MAIN 341 341 171 40 "PP_FUSED_MASK=1"
} 2>&1 | tee gpurun_out/r4/kbench5.txt
