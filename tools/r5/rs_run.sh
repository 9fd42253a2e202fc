#!/bin/bash
# round 5: the once-per-level gathers (resample, field up-sampling, composition) -- marching kernels against the general ones
#   tools/r5/rs_run.sh
cd "$(dirname "$0")/../.."
SB=tools/kbench/sbench
LIB=platipy_amd/csrc/libplatipy_hip.so
for rep in 1 2; do
  echo "== axis-aligned kernels (library default)"; timeout 120 $SB $LIB 512 512 256 10 2>&1 | grep -iE "resample|compose|warp"
  echo "== PP_RESAMPLE_GENERIC=1 (round 4's kernels)"; PP_RESAMPLE_GENERIC=1 timeout 120 $SB $LIB 512 512 256 10 2>&1 | grep -iE "resample|compose|warp"
done
for zc in 8 64; do echo "== PP_RS_ZCHUNK=$zc"; PP_RS_ZCHUNK=$zc timeout 120 $SB $LIB 512 512 256 10 2>&1 | grep -iE "resample|compose"; done
echo "== 341 x 341 x 171"; timeout 120 $SB $LIB 341 341 171 10 2>&1 | grep -iE "resample|compose"
echo "== 341 x 341 x 171, general"; PP_RESAMPLE_GENERIC=1 timeout 120 $SB $LIB 341 341 171 10 2>&1 | grep -iE "resample|compose"
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -k "marching or resample or compose or mask_prop" 2>&1 | tail -3
