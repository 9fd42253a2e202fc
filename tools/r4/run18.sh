#!/bin/bash
# round 4, GPU visit 18: one atlas chain, kernel durations by grid (= by pyramid level)
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/chain
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/chain -o chain -- python tools/profile_atlas.py > gpurun_out/chain/run.log 2>&1
grep "timed run" gpurun_out/chain/run.log
python tools/rocpd_by_grid.py gpurun_out/chain/chain_results.db fused2
python tools/rocpd_by_grid.py gpurun_out/chain/chain_results.db metric
python tools/rocpd_stats.py gpurun_out/chain/chain_results.db | sed -n 3,40p | cut -c1-160
