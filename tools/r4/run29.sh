#!/bin/bash
# round 4, GPU visit 29: z-chunk length at 341x341x171 and 405x405x200 with padded rows + mixed tiles (the launcher's choice first)
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
export KB_SPACING=1.5,1.5,1.5
MAIN=platipy_amd/csrc/libplatipy_hip.so
echo "== 341"; timeout 200 $KB $MAIN 341 341 171 30 "PP_FUSED_GEN=2" "PP_FUSED_ZCHUNK=25" "PP_FUSED_ZCHUNK=29" "PP_FUSED_ZCHUNK=35" "PP_FUSED_ZCHUNK=43" "PP_FUSED_ZCHUNK=57" "PP_FUSED_ZCHUNK=86" | cut -c1-200
echo "== 405"; timeout 200 $KB $MAIN 405 405 200 20 "PP_FUSED_GEN=2" "PP_FUSED_ZCHUNK=29" "PP_FUSED_ZCHUNK=34" "PP_FUSED_ZCHUNK=40" "PP_FUSED_ZCHUNK=50" "PP_FUSED_ZCHUNK=67" "PP_FUSED_ZCHUNK=100" | cut -c1-200
