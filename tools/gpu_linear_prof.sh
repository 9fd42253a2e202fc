#!/bin/bash
# rocprofv3 kernel statistics of the pipelines' two linear registrations at 512x512x256 (tools/profile_linear.py)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/lin
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/lin -o lin -- python tools/profile_linear.py > gpurun_out/lin/run.log 2>&1
grep -v "amdgpu.ids" gpurun_out/lin/run.log | tail -12
python tools/rocpd_stats.py gpurun_out/lin/lin_results.db | head -16 | cut -c1-200
