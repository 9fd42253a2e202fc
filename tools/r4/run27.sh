#!/bin/bash
# round 4, GPU visit 27: size gates for padded rows / MASK instances; chain; full-size iteration unchanged
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
MAIN=platipy_amd/csrc/libplatipy_hip.so
timeout 900 python -m pytest tests/test_kernels.py tests/test_registration.py -m gpu -x -q 2>&1 | tail -2
timeout 60 $KB $MAIN 512 512 256 30 "PP_FUSED_GEN=2" | cut -c1-215
export KB_SPACING=1.5,1.5,1.5
for size in "341 341 171 30" "171 171 85 60" "85 85 43 100" "128 128 64 100"; do timeout 60 $KB $MAIN $size "PP_FUSED_GEN=2"; done | cut -c1-215
timeout 300 python tools/r4/chain_levels.py
