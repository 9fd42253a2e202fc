#!/bin/bash
# round 3, A/B 2: generation-3 kernel A against generation 2 (same box, same run), several grids; then the barrier trace
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r3
{
for rep in 1 2; do
  timeout 120 $KB $V/head_r2.so 512 512 256 30 "PP_FUSED_SUM=1"
  timeout 120 $KB $MAIN 512 512 256 30 "PP_FUSED_A3=1" "PP_FUSED_A3=0"
done
for sz in "341 341 171 30" "171 171 86 40" "85 85 43 40" "512 512 512 10" "256 256 512 20"; do
  timeout 120 $KB $V/head_r2.so $sz "PP_FUSED_SUM=1"
  timeout 120 $KB $MAIN $sz "PP_FUSED_A3=1" "PP_FUSED_A3=0"
done
KB_SPACING=0.9766,0.9766,2.5 timeout 120 $KB $V/head_r2.so 512 512 256 20 "PP_FUSED_SUM=1"
KB_SPACING=0.9766,0.9766,2.5 timeout 120 $KB $MAIN 512 512 256 20 "PP_FUSED_A3=1" "PP_FUSED_A3=0"
timeout 120 $KB $V/trace3.so 512 512 256 6 "PP_FUSED_A3=1"
} 2>&1 | tee gpurun_out/r3/kbench_r3_3.txt
