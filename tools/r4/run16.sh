#!/bin/bash
# round 4, GPU visit 16: two-level ticket, block-count sweep
cd "$(dirname "$0")/../.."
timeout 300 python tools/r4/mv_check.py | grep -c 'rel err [0-9.e-]*e-1[5-9]\|rel err 0.0'
timeout 600 python -m pytest tests/test_linear.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -2
echo "== default"; timeout 200 python tools/profile_linear.py
for nb in 256 512 1024 2048 4096; do
  echo "== PP_METRIC_BLOCKS=$nb"; PP_METRIC_BLOCKS=$nb timeout 200 python tools/profile_linear.py
done
