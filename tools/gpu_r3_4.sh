#!/bin/bash
# line-search speculation depth: the pipelines' two linear registrations at 512x512x256, per level
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3b
for d in 4 3 2 1; do
  echo "== PP_LINE_SEARCH_SPECULATION=$d"
  PP_LINE_SEARCH_SPECULATION=$d timeout 300 python tools/profile_linear.py 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee gpurun_out/r3b/linear_speculation.txt
