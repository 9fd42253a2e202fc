#!/bin/bash
# quick perf/correctness check of the fused demons path
set +e
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "demons" 2>&1 | tail -3
for zc in ${ZCS:-auto}; do
  if [ "$zc" = "auto" ]; then unset PP_FUSED_ZCHUNK; else export PP_FUSED_ZCHUNK=$zc; fi
  echo "== zchunk $zc"
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-registration --no-atlas 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('ms_per_step %.4f  value %.0f  frac_iter %.3f' % (d['ms_per_step'], d['value'], d['roofline_iteration']['frac']))
        for k,v in d['kernels'].items(): print('   %-28s %.4f ms  %.0f GB/s' % (k, v['avg_ms'], v['achieved_GBps'] or 0))
    else: print(l)
"
done
