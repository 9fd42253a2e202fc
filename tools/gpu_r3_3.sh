#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3b
mkdir -p $OUT
for v in "17 17" "9 17" "9 9" "5 5" "17 9"; do
  set -- $v
  echo "PP_FIR_SP_MIN=$1 PP_FIR_XROW_MIN=$2"
  PP_FIR_SP_MIN=$1 PP_FIR_XROW_MIN=$2 timeout 300 tools/kbench/sbench platipy_amd/csrc/libplatipy_hip.so 512 512 256 5 2>&1 | grep -i "discrete"
done 2>&1 | tee $OUT/sbench_spmin.txt
