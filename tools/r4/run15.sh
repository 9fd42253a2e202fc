#!/bin/bash
# round 4, GPU visit 15: kernel durations of the linear stage by level (grid size)
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/lin
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/lin -o lin -- python tools/profile_linear.py > gpurun_out/lin/run.log 2>&1
tail -12 gpurun_out/lin/run.log
python tools/rocpd_by_grid.py gpurun_out/lin/lin_results.db metric
python tools/rocpd_by_grid.py gpurun_out/lin/lin_results.db sum14
