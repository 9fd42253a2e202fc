#!/bin/bash
# round 4, GPU visit 23: metric value + gradient in one launch
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_linear.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -2
for v in 0 1; do
  echo "== PP_METRIC_GRAD_ONE_LAUNCH=$v"; PP_METRIC_GRAD_ONE_LAUNCH=$v timeout 200 python tools/profile_linear.py
done
