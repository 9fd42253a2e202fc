#!/bin/bash
# round 4, GPU visit 17: occupancy bound of the lane-mapped probe kernel (spills against waves per SIMD)
cd "$(dirname "$0")/../.."
for lib in platipy_amd/csrc/libplatipy_hip.so tools/kbench/variants/mv_w6.so tools/kbench/variants/mv_w8.so; do
  echo "== $lib"; timeout 200 python tools/profile_linear.py $lib
done
