#!/bin/bash
# round 4, GPU visit 32: run labelling with a wavefront per row against a block per row (process_probability_image in the chain)
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_kernels.py tests/test_fusion.py tests/test_multiatlas.py -m gpu -x -q 2>&1 | tail -2
for v in block wave; do
  echo "== rows by $v"
  if [ $v == block ]; then export PP_CC_ROWS_BLOCK=1; else unset PP_CC_ROWS_BLOCK; fi
  timeout 300 python tools/r4/chain_stages.py | grep -E "process_probability|timed run"
done
