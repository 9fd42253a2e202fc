#!/bin/bash
# round 5, GPU visit 1: (a) does the tree's build still match round 4's (LDS read-slot bias), (b) drift of the blocks of an XCD,
# (c) soft synchronisation at lags 0..4, (d) no-SLP build, (e) HBM fetch of main vs sync, (f) power + clock per ablation
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
OUT=gpurun_out/r5/g1
mkdir -p $OUT
{
echo "== timing, 2 rounds"
for rep in 1 2; do
  for lib in $V/r4.so $MAIN $V/noslp.so; do timeout 120 $KB $lib 512 512 256 60 "PP_FUSED_GEN=2" | cut -c1-220; done
  timeout 200 $KB $V/sync.so 512 512 256 60 "PP_FUSED_SYNC=0" "PP_FUSED_SYNC=1" "PP_FUSED_SYNC=2" "PP_FUSED_SYNC=3" "PP_FUSED_SYNC=4" | grep -v "^drift\|^  xcd" | cut -c1-220
done
echo "== drift without / with synchronisation"
timeout 120 $KB $V/drift.so 512 512 256 20 "PP_FUSED_GEN=2" | cut -c1-220
timeout 120 $KB $V/sync.so 512 512 256 20 "PP_FUSED_SYNC=2" | cut -c1-220
timeout 120 $KB $V/sync.so 512 512 256 20 "PP_FUSED_SYNC=1" | cut -c1-220
echo "== 341 level"
export KB_SPACING=1.5,1.5,1.5
for lib in $V/r4.so $MAIN; do timeout 120 $KB $lib 341 341 171 60 "PP_FUSED_GEN=2" | cut -c1-220; done
timeout 200 $KB $V/sync.so 341 341 171 60 "PP_FUSED_SYNC=0" "PP_FUSED_SYNC=2" "PP_FUSED_SYNC=3" | grep -v "^drift\|^  xcd" | cut -c1-220
unset KB_SPACING
} 2>&1 | tee $OUT/timing.txt
{
echo "== HBM fetch / write per launch (FETCH_SIZE in 64-B units x 2 on gfx950 per round 2-4's calibration)"
for tag in main sync2 sync1; do
  case $tag in main) lib=$MAIN; env="PP_FUSED_GEN=2";; sync2) lib=$V/sync.so; env="PP_FUSED_SYNC=2";; sync1) lib=$V/sync.so; env="PP_FUSED_SYNC=1";; esac
  rm -rf $OUT/pmc_$tag
  timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_$tag -o fetch -- $KB $lib 512 512 256 6 "$env" > $OUT/pmc_$tag.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_$tag/* $OUT/pmc_$tag.md > /dev/null 2>&1 || python tools/pmc_summary.py $OUT/pmc_$tag $OUT/pmc_$tag.md
  echo "-- $tag"; grep "k_fused2" $OUT/pmc_$tag.md
done
} 2>&1 | tee $OUT/fetch.txt
{
echo "== package power and clocks per variant (3000 iterations each, hwmon sampled every 50 ms)"
for tag in main ablA_nomem ablB_nomem; do
  case $tag in main) lib=$MAIN;; *) lib=$V/$tag.so;; esac
  tools/r5/power_sample.sh $tag timeout 120 $KB $lib 512 512 256 1500 "PP_FUSED_GEN=2" | cut -c1-200
done
echo "== GRBM_GUI_ACTIVE per launch (effective clock = cycles / duration)"
for tag in main ablA_nomem ablB_nomem; do
  case $tag in main) lib=$MAIN;; *) lib=$V/$tag.so;; esac
  rm -rf $OUT/clk_$tag
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/clk_$tag -o clk -- $KB $lib 512 512 256 6 "PP_FUSED_GEN=2" > $OUT/clk_$tag.log 2>&1
  python tools/r5/clk_from_pmc.py $OUT/clk_$tag $tag
done
rocm-smi --showmaxpower 2>&1 | grep -i "max"
} 2>&1 | tee $OUT/power.txt
find $OUT -name "*.csv" -size +2M -delete
