#!/bin/bash
# where does a coarse-level iteration go: kernel durations and start-to-start gaps from the kernel trace
cd "$(dirname "$0")/../.."
ROOT=$PWD
KB=$ROOT/tools/kbench/kbench
MAIN=$ROOT/platipy_amd/csrc/libplatipy_hip.so
OUT=$ROOT/gpurun_out/r3/trace10
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 100000000; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/small_$v -- $KB $MAIN 85 85 43 100 "PP_FUSED_SMALL=$v" > $OUT/run_$v.txt 2>&1
done
cd $ROOT
python3 tools/kbench/trace_gaps.py $OUT/small_0 $OUT/small_100000000
