#!/bin/bash
# round 4, GPU visit 8: what clock and power do the fused kernels run at?  (s_memtime says ~1.7 GHz of the nominal 2.4)
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
MAIN=platipy_amd/csrc/libplatipy_hip.so
OUT=gpurun_out/r4
mkdir -p $OUT
{
echo "== idle"; rocm-smi --showpower --showclocks --showperflevel 2>&1 | grep -v "^=\|^$" | head -30
( timeout 60 $KB $MAIN 512 512 256 4000 "PP_FUSED_MASK=1" > $OUT/kbench8_long.txt 2>&1 ) &
sleep 2.0
for i in 1 2 3 4; do
  echo "== busy sample $i"; rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|socclk" | head -12
  sleep 0.5
done
wait
cat $OUT/kbench8_long.txt
echo "== rocm-smi limits"; rocm-smi --showmaxpower --showclkfrq 2>&1 | grep -v "^=\|^$" | head -60
} 2>&1 | tee $OUT/clocks8.txt
