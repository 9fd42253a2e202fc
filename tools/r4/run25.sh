#!/bin/bash
# round 4, GPU visit 25: the body / wrapper refactor must not move the full-size iteration; mixed launches in the chain
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
MAIN=platipy_amd/csrc/libplatipy_hip.so
OLD=tools/kbench/variants/g3_committed.so
for rep in 1 2 3; do
  for lib in $OLD $MAIN; do timeout 60 $KB $lib 512 512 256 30 "PP_FUSED_MASK=1"; done
done 2>&1 | cut -c1-210
timeout 600 python -m pytest tests/test_kernels.py tests/test_registration.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/r4/chain_levels.py
