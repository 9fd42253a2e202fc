#!/bin/bash
# round 5: SmoothingRecursiveGaussian of a field (three directional passes), the library against another build / the old kernels
#   tools/r5/rg_run.sh [other.so ...]
cd "$(dirname "$0")/../.."
SB=tools/kbench/sbench
for rep in 1 2; do
  for lib in platipy_amd/csrc/libplatipy_hip.so "$@"; do
    echo "== $lib"; timeout 120 $SB $lib 512 512 256 10 2>&1 | grep -i "recursive"
  done
  echo "== library, PP_RG_SEG_V1=1 (round 3's strided kernels)"; PP_RG_SEG_V1=1 timeout 120 $SB platipy_amd/csrc/libplatipy_hip.so 512 512 256 10 2>&1 | grep -i "recursive"
done
echo "== 340 x 340 x 170"; for e in "PP_X=0" "PP_RG_SEG_V1=1"; do env $e timeout 120 $SB platipy_amd/csrc/libplatipy_hip.so 340 340 170 10 2>&1 | grep -i "recursive"; done
echo "== 341 x 341 x 171 (rows that are not whole quads)"; for lib in platipy_amd/csrc/libplatipy_hip.so "$@"; do timeout 120 $SB $lib 341 341 171 10 2>&1 | grep -i "recursive"; done
timeout 600 python -m pytest tests/test_kernels.py tests/test_registration.py -m gpu -x -q -k "recursive or single_sweep" 2>&1 | tail -3
