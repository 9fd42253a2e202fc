#!/bin/bash
# fused-schedule throughput vs grid size: tile waste and row alignment
cd "$(dirname "$0")/.."
for sz in "384 352 171" "341 341 171" "344 341 171" "352 352 171" "320 336 171" "512 512 256"; do
  python bench.py --size $sz --steps 40 --warmup 5 --no-cpu-baseline --no-registration --no-atlas 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('size $sz: ms/iter %.4f  Mvox/s %.0f' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
