#!/bin/bash
# metric-probe latency by block cap (tools/bench_metric.py), then the linear stage per level and the bench's atlas legs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3b
{
echo "== default caps"
timeout 300 python tools/bench_metric.py 2>&1 | grep -v amdgpu.ids | grep "batch"
for nb in 1024 256; do
  echo "== PP_METRIC_BLOCKS=$nb"
  PP_METRIC_BLOCKS=$nb timeout 300 python tools/bench_metric.py 2>&1 | grep -v amdgpu.ids | grep "batch"
done
} 2>&1 | tee gpurun_out/r3b/metric_blocks2.txt
bash tools/gpu_r3_5.sh
