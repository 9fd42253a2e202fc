#!/bin/bash
# round 3, GPU call 1: instruction-rate probe, the new full-size oracle parity tests, then the whole -m gpu suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
timeout 120 tools/probes/valu_rate > gpurun_out/r3/valu_rate.txt 2>&1
cat gpurun_out/r3/valu_rate.txt
export PP_STATS_DIR=gpurun_out/r3/parity_stats
timeout 1500 python -m pytest tests/test_fullsize_oracle.py -x -q -m gpu -s 2>&1 | tail -40 | tee gpurun_out/r3/fullsize_oracle.log
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_fullsize_oracle.py 2>&1 | tail -8 | tee gpurun_out/r3/gpu_suite.log
