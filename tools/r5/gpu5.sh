#!/bin/bash
# round 5, GPU visit 5: kernel A's packed-pair ESM gradients (main) against the scalar statements (esm0) and the pair-held z
# window (zpairs): time, checksums, VALU instruction counters; soft synchronisation that works (agent-scope reads, plain
# stores) at lags 2..32 with the HBM fetch beside the time
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
OUT=gpurun_out/r5/g5
mkdir -p $OUT
{
echo "== timing (3 rounds, alternating)"
for rep in 1 2 3; do
  for lib in $V/r4.so $MAIN $V/esm0.so $V/zpairs.so; do timeout 120 $KB $lib 512 512 256 60 "PP_FUSED_GEN=2" | cut -c1-220; done
done
echo "== soft synchronisation (agent-scope reads of the progress words, plain stores)"
for rep in 1 2; do
  timeout 300 $KB $V/syncaw.so 512 512 256 40 "PP_FUSED_SYNC=0" "PP_FUSED_SYNC=2" "PP_FUSED_SYNC=4" "PP_FUSED_SYNC=8" "PP_FUSED_SYNC=16" "PP_FUSED_SYNC=32" | cut -c1-220
done
echo "== 341 level"
export KB_SPACING=1.5,1.5,1.5
for lib in $V/r4.so $MAIN; do timeout 120 $KB $lib 341 341 171 60 "PP_FUSED_GEN=2" | cut -c1-220; done
timeout 200 $KB $V/syncaw.so 341 341 171 60 "PP_FUSED_SYNC=0" "PP_FUSED_SYNC=4" "PP_FUSED_SYNC=16" | cut -c1-220
unset KB_SPACING
} 2>&1 | tee $OUT/timing.txt
{
echo "== FETCH_SIZE per launch (unit 2048 B by the calibration copies)"
for tag in sync0 sync4 sync16; do
  case $tag in sync0) env="PP_FUSED_SYNC=0";; sync4) env="PP_FUSED_SYNC=4";; sync16) env="PP_FUSED_SYNC=16";; esac
  rm -rf $OUT/pmc_$tag
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_$tag -o fetch -- $KB $V/syncaw.so 512 512 256 6 "$env" > $OUT/pmc_$tag.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_$tag $OUT/pmc_$tag.md > /dev/null 2>&1
  echo "-- $tag"; grep "k_fused2" $OUT/pmc_$tag.md
done
echo "== SQ_INSTS_VALU / SALU / LDS, SQ_WAVE_CYCLES, SQ_WAIT_ANY per launch: r4 against main"
for tag in r4 main; do
  case $tag in r4) lib=$V/r4.so;; main) lib=$MAIN;; esac
  rm -rf $OUT/sq_$tag
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/sq_$tag -o sq -- $KB $lib 512 512 256 6 "PP_FUSED_GEN=2" > $OUT/sq_$tag.log 2>&1
  python tools/pmc_summary.py $OUT/sq_$tag $OUT/sq_$tag.md > /dev/null 2>&1
  echo "-- $tag"; grep "k_fused2_force" $OUT/sq_$tag.md
done
} 2>&1 | tee $OUT/counters.txt
find $OUT -name "*.csv" -size +6M -delete
