#!/bin/bash
# rocprofv3 kernel statistics of one multi-atlas chain (bench.py's multi_atlas leg shape)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/chain
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/chain -o chain -- python tools/profile_atlas.py > gpurun_out/chain/run.log 2>&1
python tools/rocpd_stats.py gpurun_out/chain/chain_results.db | head -45 | cut -c1-200
