#!/bin/bash
# round 4, GPU visit 22: block count and speculation depth with the pipelined probe kernel
cd "$(dirname "$0")/../.."
for nb in 512 1024 2048 4096; do
  echo "== PP_METRIC_BLOCKS=$nb"; PP_METRIC_BLOCKS=$nb timeout 200 python tools/profile_linear.py | grep -v "^quick\|^affine" 
done
for d in 3 2; do
  echo "== PP_LINE_SEARCH_SPECULATION=$d"; PP_LINE_SEARCH_SPECULATION=$d timeout 200 python tools/profile_linear.py | grep -v "^quick\|^affine"
done
