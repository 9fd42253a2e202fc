#!/bin/bash
# round 4, GPU visit 7: kernel A's LDS bank conflicts (padded image-tile pitch, permuted x-pass lanes): time and counters
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
KB=$PWD/tools/kbench/kbench
V=$PWD/tools/kbench/variants
MAIN=$PWD/platipy_amd/csrc/libplatipy_hip.so
OUT=$PWD/gpurun_out/r4
mkdir -p $OUT
{
for rep in 1 2 3; do
  for lib in $V/lds_old.so $V/lds_pad.so $V/lds_perm.so $MAIN; do
    timeout 60 $KB $lib 512 512 256 30 "PP_FUSED_MASK=1"
  done
done
timeout 60 $KB $MAIN 340 340 170 40 "PP_FUSED_MASK=1"
timeout 60 $KB $V/lds_old.so 340 340 170 40 "PP_FUSED_MASK=1"
} 2>&1 | tee $OUT/kbench7.txt
for lib in lds_old main; do
  L=$V/$lib.so; [ $lib == main ] && L=$MAIN
  ( cd /tmp && timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc7_$lib -o c -- $KB $L 512 512 256 6 "PP_FUSED_MASK=1" > $OUT/pmc7_$lib.log 2>&1 )
  python tools/pmc_summary.py $OUT/pmc7_$lib $OUT/pmc7_$lib.md > /dev/null 2>&1
  grep "k_fused2" $OUT/pmc7_$lib.md
done
