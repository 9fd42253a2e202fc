#!/bin/bash
# round 5, GPU visit 2: soft synchronisation through the XCD's own L2 (workgroup-scope peek + atomic), lags 1..8: time, drift,
# HBM fetch; then the new -m gpu tests (RCCL world 1, oriented filter seam)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
OUT=gpurun_out/r5/g2
mkdir -p $OUT
{
echo "== timing"
for rep in 1 2; do
  timeout 120 $KB $MAIN 512 512 256 40 "PP_FUSED_GEN=2" | cut -c1-220
  timeout 300 $KB $V/syncnd.so 512 512 256 40 "PP_FUSED_SYNC=0" "PP_FUSED_SYNC=1" "PP_FUSED_SYNC=2" "PP_FUSED_SYNC=3" "PP_FUSED_SYNC=4" "PP_FUSED_SYNC=6" "PP_FUSED_SYNC=8" "PP_FUSED_SYNC=12" "PP_FUSED_SYNC=20" | cut -c1-220
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
echo "== FETCH_SIZE per launch (x 2 x 64 B on gfx950 per the calibration of rounds 2-4: value * 64 * 2 bytes?  raw values below)"
for tag in main sync2 sync4 sync8; do
  case $tag in main) lib=$MAIN; env="PP_FUSED_GEN=2";; sync2) lib=$V/syncnd.so; env="PP_FUSED_SYNC=2";; sync4) lib=$V/syncnd.so; env="PP_FUSED_SYNC=4";; sync8) lib=$V/syncnd.so; env="PP_FUSED_SYNC=8";; esac
  rm -rf $OUT/pmc_$tag
  KB_CALIBRATE=1 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_$tag -o fetch -- $KB $lib 512 512 256 6 "$env" > $OUT/pmc_$tag.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_$tag $OUT/pmc_$tag.md > /dev/null 2>&1
  echo "-- $tag"; grep "k_fused2\|k_cal" $OUT/pmc_$tag.md
done
} 2>&1 | tee $OUT/fetch.txt
{
echo "== new GPU tests"
timeout 900 python -m pytest tests/test_rccl_world1.py tests/test_sitk_seam.py -m gpu -x -q 2>&1 | tail -15
} 2>&1 | tee $OUT/tests.txt
find $OUT -name "*.csv" -size +4M -delete
