#!/bin/bash
# rocprofv3 kernel statistics of config 2's whole registration (two runs: warm-up + timed)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/reg
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/reg -o reg -- python tools/profile_registration.py > gpurun_out/reg/run.log 2>&1
grep registration_s gpurun_out/reg/run.log
python tools/rocpd_stats.py gpurun_out/reg/reg_results.db | head -34 | cut -c1-175
