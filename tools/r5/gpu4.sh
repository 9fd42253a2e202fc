#!/bin/bash
# round 5, GPU visit 4: (a) which XCD does block b run on (HW_REG_XCC_ID)?  (b) which scope pair makes the progress words
# visible, and at what cost (load scope x store scope, lag 2)?  (c) 4 chains on 4 streams with GPU_MAX_HW_QUEUES=8
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
KB=tools/kbench/kbench
V=tools/kbench/variants
OUT=gpurun_out/r5/g4
mkdir -p $OUT
{
for v in ww aw wa aa; do
  echo "== load/store scope $v (w = workgroup: sc0, a = agent: sc1), lag 2 then lag 0"
  timeout 120 $KB $V/sync_$v.so 512 512 256 20 "PP_FUSED_SYNC=2" | grep -v "xcd [1-6]" | cut -c1-220
  timeout 120 $KB $V/sync_$v.so 512 512 256 20 "PP_FUSED_SYNC=0" | grep -v "xcd\|^drift\|xcc\|blocks 0" | cut -c1-220
done
} 2>&1 | tee $OUT/scopes.txt
{
echo "== 4 chains on 4 streams, GPU_MAX_HW_QUEUES unset -> platipy_amd sets 8"
rm -rf $OUT/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python tools/r5/streams_timeline.py run 4 4 2>&1 | grep TIMELINE_RUN
python tools/r5/streams_timeline.py analyse $OUT/tl > $OUT/streams_timeline.md
cat $OUT/streams_timeline.md
echo "== without the profiler: 4 hardware queues against 8"
for q in 4 8; do GPU_MAX_HW_QUEUES=$q timeout 300 python tools/r5/streams_timeline.py run 4 4 2>&1 | grep TIMELINE_RUN | sed "s/^/GPU_MAX_HW_QUEUES=$q: /"; done
} 2>&1 | tee $OUT/timeline.txt
find $OUT -name "*.csv" -size +6M -delete
