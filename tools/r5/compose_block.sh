#!/bin/bash
# round 5: the composition's block shape (64 x 4 default | 128 x 2 | 256 x 1) with the banded and the plain launch
cd "$(dirname "$0")/../.."
SB=tools/kbench/sbench; LIB=platipy_amd/csrc/libplatipy_hip.so
for size in "512 512 256" "341 341 171"; do
  for b in 64 128 256; do for band in 1 0; do echo "== $size PP_COMPOSE_BLOCK=$b PP_RS_BAND=$band"; PP_COMPOSE_BLOCK=$b PP_RS_BAND=$band timeout 120 $SB $LIB $size 10 2>&1 | grep -iE "compose"; done; done
done
for b in 64 256; do echo "== registration PP_COMPOSE_BLOCK=$b"; for i in 1 2 3; do PP_COMPOSE_BLOCK=$b python tools/profile_registration.py 2>/dev/null | grep registration_s; done; done
timeout 900 python -m pytest tests/test_kernels.py tests/test_registration.py -m gpu -x -q -k "banded or compose or registration_matches" 2>&1 | tail -3
