#!/bin/bash
# round 4, GPU visit 12: fused Gaussian with the plane prefetched one step ahead and a masked b128 store, against the committed one
cd "$(dirname "$0")/../.."
SB=tools/kbench/sbench
MAIN=platipy_amd/csrc/libplatipy_hip.so
OLD=tools/kbench/variants/g3_committed.so
for rep in 1 2; do
  for lib in $OLD $MAIN; do
    echo "== $lib"; timeout 90 $SB $lib 512 512 256 20 2>&1 | grep -i "gauss"
  done
done
echo "== 3 passes"; PP_GAUSS3=0 timeout 90 $SB $MAIN 512 512 256 20 2>&1 | grep -i "gauss"
timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "gaussian or fir or smooth" 2>&1 | tail -3
