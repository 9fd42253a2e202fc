#!/bin/bash
# round 4, GPU visit 30: the fused kernels without the SLP vectoriser (v_fma_f32 instead of v_pk_fma_f32 + operand moves)
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
MAIN=platipy_amd/csrc/libplatipy_hip.so
V=tools/kbench/variants
for rep in 1 2 3; do
  for lib in $MAIN $V/noslp.so $V/novec.so; do timeout 60 $KB $lib 512 512 256 30 "PP_FUSED_GEN=2"; done
done 2>&1 | cut -c1-200
export KB_SPACING=1.5,1.5,1.5
for rep in 1 2; do
  for lib in $MAIN $V/noslp.so; do timeout 60 $KB $lib 341 341 171 30 "PP_FUSED_GEN=2"; done
done 2>&1 | cut -c1-200
