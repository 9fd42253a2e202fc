#!/bin/bash
# round 4, GPU visit 20: lanes without a candidate idle; speculation depth per level
cd "$(dirname "$0")/../.."
timeout 300 python tools/r4/mv_check.py | grep -c 'rel err [0-9.e-]*e-1[5-9]\|rel err 0.0'
timeout 600 python -m pytest tests/test_linear.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -2
for d in 4 3 2 1; do
  echo "== PP_LINE_SEARCH_SPECULATION=$d"; PP_LINE_SEARCH_SPECULATION=$d timeout 200 python tools/profile_linear.py
done
