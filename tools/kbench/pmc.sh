#!/bin/bash
# One rocprofv3 --pmc pass per counter set over tools/kbench/kbench for a build of the library; mean per launch of the two fused
# kernels (tools/pmc_summary.py).  FETCH_SIZE and WRITE_SIZE need passes of their own on gfx950 ("Request exceeds the
# capabilities of the hardware" otherwise); KB_CALIBRATE=1 adds two copies with known byte counts (268 435 456 bytes read and
# written each at 512 x 512 x 256: FETCH_SIZE 131 085 -> the unit is 2 KB).
#   tools/kbench/pmc.sh <out dir> <build[:ENV=V]> <set> [<set> ...]      set = counters separated by commas, or a name:
#     fetch = FETCH_SIZE | write = WRITE_SIZE | l2 = TCC_HIT_sum,TCC_MISS_sum | clock = GRBM_GUI_ACTIVE
#     issue = SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_ACTIVE_INST_VALU
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=$1; spec=$2; shift 2
name=${spec%%:*}
case $name in main) lib=platipy_amd/csrc/libplatipy_hip.so;; */*|*.so) lib=$name;; *) lib=tools/kbench/variants/$name.so;; esac
env="PP_FUSED_GEN=2"; [ "$spec" != "$name" ] && env=${spec#*:}
mkdir -p $OUT
for set in "$@"; do
  case $set in
    fetch) c="FETCH_SIZE";; write) c="WRITE_SIZE";; l2) c="TCC_HIT_sum TCC_MISS_sum";; clock) c="GRBM_GUI_ACTIVE";;
    issue) c="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU";;
    *) c=${set//,/ };;
  esac
  tag=${name}_$(echo $set | tr ',' '_')
  rm -rf $OUT/$tag
  KB_CALIBRATE=${KB_CALIBRATE:-1} timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- tools/kbench/kbench $lib ${KB_SIZE:-512 512 256} 6 "$env" > $OUT/$tag.log 2>&1
  python tools/pmc_summary.py $OUT/$tag $OUT/$tag.md > /dev/null 2>&1
  echo "-- $name ($env): $c"; grep "k_fused2\|k_cal" $OUT/$tag.md
  [ "$set" = clock ] && python tools/r5/clk_from_pmc.py $OUT/$tag $name
done
find $OUT -name "*.csv" -size +6M -delete
