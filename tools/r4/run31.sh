#!/bin/bash
# round 4, GPU visit 31: compiler scheduling options on the fused kernels
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
MAIN=platipy_amd/csrc/libplatipy_hip.so
V=tools/kbench/variants
for rep in 1 2; do
  for lib in $MAIN $V/relocc.so $V/nomisched.so $V/O2.so; do timeout 60 $KB $lib 512 512 256 30 "PP_FUSED_GEN=2"; done
done 2>&1 | cut -c1-200
