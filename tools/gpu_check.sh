#!/bin/bash
# One GPU-box visit: GPU parity tests, smoke, bench (fused + staged), z-chunk sweep, rocprofv3 stats.
# Everything lands in gpurun_out/ (merged back by gpurun).
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench fused"; timeout 600 python bench.py --steps 30 --warmup 3 2>gpurun_out/bench_fused.err | tee gpurun_out/bench_fused.json
echo "== bench staged"; timeout 300 python bench.py --variant staged --steps 10 --warmup 2 --no-cpu-baseline --no-registration --no-atlas 2>gpurun_out/bench_staged.err | tee gpurun_out/bench_staged.json
for zc in 8 16 64; do
  echo "== zchunk $zc"; PP_FUSED_ZCHUNK=$zc timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-registration --no-atlas 2>/dev/null | tee gpurun_out/bench_zc$zc.json
done
echo "== rocprofv3"; timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r1 -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-registration --no-atlas > gpurun_out/prof_run.log 2>&1
find gpurun_out/prof -name "*stats*" | head
tail -3 gpurun_out/prof_run.log
