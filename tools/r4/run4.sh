#!/bin/bash
# round 4, GPU visit 4: kernel B's x pass + strip loads at the end of the plane step (a whole step of prefetch distance).
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r4
{
for rep in 1 2 3; do
  for lib in $V/noxlate.so $MAIN $V/xlate_novote.so $V/xlate_finbefore.so; do
    timeout 120 $KB $lib 512 512 256 30 "PP_FUSED_MASK=1"
  done
done
timeout 120 $KB $MAIN 340 340 170 40 "PP_FUSED_MASK=1"
timeout 120 $KB $V/noxlate.so 340 340 170 40 "PP_FUSED_MASK=1"
timeout 120 $KB $MAIN 341 341 171 40 "PP_FUSED_MASK=1"
timeout 120 $KB $V/noxlate.so 341 341 171 40 "PP_FUSED_MASK=1"
} 2>&1 | tee gpurun_out/r4/kbench4.txt
