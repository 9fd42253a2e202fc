#!/bin/bash
# round 4, GPU visit 10: kernel B's HBM fetch on the BENCH's data by build variant (FETCH_SIZE of bench.py under rocprofv3)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
ROOT=$PWD
V=$PWD/tools/kbench/variants
OUT=$PWD/gpurun_out/r4
mkdir -p $OUT
cp platipy_amd/csrc/libplatipy_hip.so /tmp/main.so
BENCH="python bench.py --steps 10 --warmup 2 --repeats 1 --no-registration --no-atlas --no-cpu-baseline"
for spec in "main:/tmp/main.so:1" "mask0:/tmp/main.so:0" "noxlate:$V/noxlate.so:1" "novoteB:$V/novoteB.so:1" "noxlate_novote:$V/noxlate_novote.so:1" "noxlate_novote_mask0:$V/noxlate_novote.so:0"; do
  name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; mask=${rest#*:}
  cp $lib platipy_amd/csrc/libplatipy_hip.so
  ( cd /tmp && PP_FUSED_MASK=$mask timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc10_$name -o c -- bash -c "cd $ROOT && $BENCH" > $OUT/pmc10_$name.log 2>&1 )
  python tools/pmc_summary.py $OUT/pmc10_$name $OUT/pmc10_$name.md > /dev/null 2>&1
  echo "== $name (PP_FUSED_MASK=$mask)"; grep "k_fused2.*FETCH" $OUT/pmc10_$name.md
done
cp /tmp/main.so platipy_amd/csrc/libplatipy_hip.so
