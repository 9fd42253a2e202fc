#!/bin/bash
# round 4, GPU visit 9: kernel B's HBM fetch rose from 1.25 GB (round 3) to 1.51 GB per launch -- which change did it?
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
KB=$PWD/tools/kbench/kbench
V=$PWD/tools/kbench/variants
MAIN=$PWD/platipy_amd/csrc/libplatipy_hip.so
OUT=$PWD/gpurun_out/r4
mkdir -p $OUT
for spec in "main:$MAIN:PP_FUSED_MASK=1" "mask0:$MAIN:PP_FUSED_MASK=0" "noxlate:$V/noxlate.so:PP_FUSED_MASK=1" "novoteB:$V/novoteB.so:PP_FUSED_MASK=1" "zchunk64:$MAIN:PP_FUSED_MASK=1,PP_FUSED_ZCHUNK=64"; do
  name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; env=${rest#*:}
  ( cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc9_$name -o c -- $KB $lib 512 512 256 6 "$env" > $OUT/pmc9_$name.log 2>&1 )
  python tools/pmc_summary.py $OUT/pmc9_$name $OUT/pmc9_$name.md > /dev/null 2>&1
  echo "== $name ($env)"; grep "k_fused2.*FETCH" $OUT/pmc9_$name.md; grep "ms/iter" $OUT/pmc9_$name.log | cut -c40-200
done
