#!/bin/bash
# 64x16 vs 32x32 tiles of the fused kernels across grid sizes (PP_FUSED_TILE forces the shape; auto = cost model)
cd "$(dirname "$0")/.."
for sz in "341 341 171" "384 352 171" "171 171 85" "85 85 43" "512 512 256" "256 256 128" "300 200 150"; do
  for tile in 0 1 auto; do
    if [ "$tile" = "auto" ]; then unset PP_FUSED_TILE; else export PP_FUSED_TILE=$tile; fi
    python bench.py --size $sz --steps 40 --warmup 5 --no-cpu-baseline --no-registration --no-atlas 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('size $sz tile $tile: ms/iter %.4f  Mvox/s %.0f' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
  done
done
