#!/bin/bash
# round 4, GPU visit 13: line-search value probes with a candidate per LANE against a candidate loop per thread
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_linear.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3
for v in 0 1; do
  echo "== PP_METRIC_LANES=$v"; PP_METRIC_LANES=$v timeout 200 python tools/profile_linear.py
done
for nb in 512 2048; do
  echo "== PP_METRIC_LANES=1 PP_METRIC_BLOCKS=$nb"; PP_METRIC_BLOCKS=$nb timeout 200 python tools/profile_linear.py
done
