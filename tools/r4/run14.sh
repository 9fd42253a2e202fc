#!/bin/bash
# round 4, GPU visit 14: lane-mapped probes without per-block fences
cd "$(dirname "$0")/../.."
timeout 300 python tools/r4/mv_check.py | grep -c 'rel err [0-9.e-]*e-1[5-9]\|rel err 0.0'
timeout 600 python -m pytest tests/test_linear.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -2
for v in 0 1; do
  echo "== PP_METRIC_LANES=$v"; PP_METRIC_LANES=$v timeout 200 python tools/profile_linear.py
done
for nb in 512 2048 4096; do
  echo "== PP_METRIC_LANES=1 PP_METRIC_BLOCKS=$nb"; PP_METRIC_BLOCKS=$nb timeout 200 python tools/profile_linear.py
done
