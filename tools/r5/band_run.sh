#!/bin/bash
# round 5: banded (XCD-aware) launch of the gathers through a field against the plain 3-D grid (PP_RS_BAND=0)
cd "$(dirname "$0")/../.."
SB=tools/kbench/sbench
LIB=platipy_amd/csrc/libplatipy_hip.so
for rep in 1 2; do
  echo "== banded (default)"; timeout 120 $SB $LIB 512 512 256 10 2>&1 | grep -iE "resample|compose|warp"
  echo "== PP_RS_BAND=0"; PP_RS_BAND=0 timeout 120 $SB $LIB 512 512 256 10 2>&1 | grep -iE "resample|compose|warp"
done
echo "== 341 x 341 x 171 banded"; timeout 120 $SB $LIB 341 341 171 10 2>&1 | grep -iE "resample|compose|warp"
echo "== 341 x 341 x 171 PP_RS_BAND=0"; PP_RS_BAND=0 timeout 120 $SB $LIB 341 341 171 10 2>&1 | grep -iE "resample|compose|warp"
for i in 1 2 3; do python tools/profile_registration.py 2>/dev/null | grep registration_s; done
echo "== PP_RS_BAND=0"; for i in 1 2 3; do PP_RS_BAND=0 python tools/profile_registration.py 2>/dev/null | grep registration_s; done
bash tools/r5/reg_pmc.sh | head -12
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -k "banded or marching or resample or compose or mask_prop or warp" 2>&1 | tail -3
