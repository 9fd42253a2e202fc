#!/bin/bash
# round 5, GPU visit 3: soft synchronisation with per-block progress words (plain stores + L1-bypassing loads), lags 1..8:
# time, drift, fetch; the RCCL world-1 tests; the 4-stream kernel timeline
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
OUT=gpurun_out/r5/g3
mkdir -p $OUT
{
echo "== timing"
for rep in 1 2; do
  timeout 120 $KB $MAIN 512 512 256 40 "PP_FUSED_GEN=2" | cut -c1-220
  timeout 300 $KB $V/syncnd.so 512 512 256 40 "PP_FUSED_SYNC=0" "PP_FUSED_SYNC=1" "PP_FUSED_SYNC=2" "PP_FUSED_SYNC=3" "PP_FUSED_SYNC=4" "PP_FUSED_SYNC=6" "PP_FUSED_SYNC=8" "PP_FUSED_SYNC=16" | cut -c1-220
done
echo "== drift with synchronisation"
timeout 120 $KB $V/sync.so 512 512 256 20 "PP_FUSED_SYNC=2" | grep -v "xcd [1-6]" | cut -c1-220
timeout 120 $KB $V/sync.so 512 512 256 20 "PP_FUSED_SYNC=6" | grep -v "xcd [1-6]" | cut -c1-220
echo "== 341 level"
export KB_SPACING=1.5,1.5,1.5
timeout 120 $KB $MAIN 341 341 171 60 "PP_FUSED_GEN=2" | cut -c1-220
timeout 200 $KB $V/syncnd.so 341 341 171 60 "PP_FUSED_SYNC=0" "PP_FUSED_SYNC=2" "PP_FUSED_SYNC=4" "PP_FUSED_SYNC=8" | cut -c1-220
unset KB_SPACING
} 2>&1 | tee $OUT/timing.txt
{
echo "== FETCH_SIZE per launch (unit 2048 B by the calibration copies: 268 435 456 bytes read = 131 085)"
for tag in main sync2 sync4; do
  case $tag in main) lib=$MAIN; env="PP_FUSED_GEN=2";; sync2) lib=$V/syncnd.so; env="PP_FUSED_SYNC=2";; sync4) lib=$V/syncnd.so; env="PP_FUSED_SYNC=4";; esac
  rm -rf $OUT/pmc_$tag
  KB_CALIBRATE=1 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_$tag -o fetch -- $KB $lib 512 512 256 6 "$env" > $OUT/pmc_$tag.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_$tag $OUT/pmc_$tag.md > /dev/null 2>&1
  echo "-- $tag"; grep "k_fused2\|k_cal" $OUT/pmc_$tag.md
done
} 2>&1 | tee $OUT/fetch.txt
{
echo "== new GPU tests"
timeout 900 python -m pytest tests/test_rccl_world1.py -m gpu -x -q 2>&1 | tail -15
} 2>&1 | tee $OUT/tests.txt
{
echo "== 4 chains on 4 streams under rocprofv3 --kernel-trace"
rm -rf $OUT/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python tools/r5/streams_timeline.py run 4 4 2>&1 | grep TIMELINE_RUN
python tools/r5/streams_timeline.py analyse $OUT/tl > $OUT/streams_timeline.md
cat $OUT/streams_timeline.md
} 2>&1 | tee $OUT/timeline.txt
find $OUT -name "*.csv" -size +6M -delete
