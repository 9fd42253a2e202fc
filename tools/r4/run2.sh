#!/bin/bash
# round 4, GPU visit 2: where a plane step of the MASK kernels goes (barrier trace, ablations of kernel B) and what makes
# a v_cndmask slow (mask-source probes).
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r4
{
for rep in 1 2; do
  timeout 120 $KB $MAIN 512 512 256 30 "PP_FUSED_MASK=1"
  for lib in $V/abl_nogather.so $V/abl_nostore.so $V/abl_noload.so; do
    timeout 120 $KB $lib 512 512 256 30 "PP_FUSED_MASK=1"
  done
done
timeout 120 $KB $V/trace.so 512 512 256 12 "PP_FUSED_MASK=1"
} 2>&1 | tee gpurun_out/r4/kbench2.txt
timeout 300 tools/probes/valu_rate 2>&1 | tee gpurun_out/r4/valu_rate2.txt
