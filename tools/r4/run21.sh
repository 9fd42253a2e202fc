#!/bin/bash
# round 4, GPU visit 21: where a probe's time goes (no sample loop / no fold / neither)
cd "$(dirname "$0")/../.."
for lib in platipy_amd/csrc/libplatipy_hip.so tools/kbench/variants/mv_noloop.so tools/kbench/variants/mv_nofold.so tools/kbench/variants/mv_neither.so; do
  echo "== $lib"; timeout 200 python tools/r4/mv_time.py $lib
done
