#!/bin/bash
# round-end evidence: GPU tests, smoke, bench (fused default, staged), rocprofv3 kernel stats, PMC passes
set +e
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rm -rf gpurun_out/final; mkdir -p gpurun_out/final
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/final/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/final/smoke.log
echo "== bench"; timeout 900 python bench.py 2>gpurun_out/final/bench.err | tee gpurun_out/final/bench.json | cut -c1-400
echo "== bench staged"; timeout 300 python bench.py --variant staged --steps 10 --warmup 2 --no-cpu-baseline --no-registration --no-atlas 2>/dev/null | tee gpurun_out/final/bench_staged.json | cut -c1-200
echo "== rocprofv3 stats (same command as the bench line, fewer steps)"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/final/prof -o bench -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-registration --no-atlas > gpurun_out/final/prof.log 2>&1
grep '^{"metric"' gpurun_out/final/prof.log > gpurun_out/final/bench_under_rocprof.json   # the bench line of the profiled run itself
tail -1 gpurun_out/final/prof.log | cut -c1-200
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/final/pmc -o $tag -- python tools/pmc_probe.py > gpurun_out/final/pmc_$tag.log 2>&1
done
ls gpurun_out/final gpurun_out/final/prof gpurun_out/final/pmc | head -40
