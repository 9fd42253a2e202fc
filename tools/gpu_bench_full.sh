#!/bin/bash
set +e
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
echo "== 2-rank run on one GPU is not possible; profile of the registration + atlas legs"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_full -o full -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_full.log 2>&1
tail -2 gpurun_out/prof_full.log
ls gpurun_out/prof_full | head
