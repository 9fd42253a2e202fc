#!/bin/bash
# z-chunk sweep of the fused schedule on the pipelines' finest isotropic grid (341x341x171 at 1.5 mm)
cd "$(dirname "$0")/.."
for zc in auto 86 57 43 34 28 24 19 14; do
  if [ "$zc" = "auto" ]; then unset PP_FUSED_ZCHUNK; else export PP_FUSED_ZCHUNK=$zc; fi
  python bench.py --size 341 341 171 --steps 40 --warmup 5 --no-cpu-baseline --no-registration --no-atlas 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('zchunk $zc: ms/iter %.4f' % d['ms_per_step'], {k: round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
