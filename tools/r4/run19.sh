#!/bin/bash
# round 4, GPU visit 19: z-chunk length x tile shape at the pipelines' demons levels (341x341x171, 171x171x85, 85x85x43), sigma as the pipeline's
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
export KB_SPACING=1.5,1.5,1.5
MAIN=platipy_amd/csrc/libplatipy_hip.so
echo "== 341x341x171"
timeout 60 $KB $MAIN 341 341 171 30 "PP_FUSED_MASK=1"
for t in 0 1; do for zc in 22 25 29 35 43 57 86 171; do
  timeout 60 $KB $MAIN 341 341 171 30 "PP_FUSED_TILE=$t,PP_FUSED_ZCHUNK=$zc"
done; done
echo "== 171x171x85"
timeout 60 $KB $MAIN 171 171 85 60 "PP_FUSED_MASK=1"
for t in 0 1; do for zc in 4 6 8 11 15 22 29 43; do
  timeout 60 $KB $MAIN 171 171 85 60 "PP_FUSED_TILE=$t,PP_FUSED_ZCHUNK=$zc"
done; done
echo "== 85x85x43"
timeout 60 $KB $MAIN 85 85 43 100 "PP_FUSED_MASK=1"
for t in 0 1; do for zc in 2 3 4 6 8 11 15 22; do
  timeout 60 $KB $MAIN 85 85 43 100 "PP_FUSED_TILE=$t,PP_FUSED_ZCHUNK=$zc"
done; done
