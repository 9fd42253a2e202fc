#!/bin/bash
# round 5, GPU visit 6: wave roles reversed in every other block (kernel A: ESM rounds / x-pass items, kernel B: strip rows):
# main (both) against noflip / flipA / flipB and round 4's library; the new -m gpu tests
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
OUT=gpurun_out/r5/g6
mkdir -p $OUT
{
echo "== timing (3 rounds, alternating)"
for rep in 1 2 3; do
  for lib in $V/r4.so $V/noflip.so $MAIN $V/flipA.so $V/flipB.so; do timeout 120 $KB $lib 512 512 256 60 "PP_FUSED_GEN=2" | cut -c1-220; done
done
echo "== 341 level"
export KB_SPACING=1.5,1.5,1.5
for rep in 1 2; do
  for lib in $V/r4.so $V/noflip.so $MAIN; do timeout 120 $KB $lib 341 341 171 60 "PP_FUSED_GEN=2" | cut -c1-220; done
done
unset KB_SPACING
} 2>&1 | tee $OUT/timing.txt
{
echo "== new GPU tests"
timeout 1200 python -m pytest tests/test_linear.py tests/test_sitk_seam.py tests/test_abi.py -m gpu -x -q -k "itk or jitter or oriented or abi" 2>&1 | tail -8
echo "== fused demons kernel tests"
timeout 1200 python -m pytest tests/test_kernels.py -m gpu -x -q -k "demons or fused" 2>&1 | tail -4
} 2>&1 | tee $OUT/tests.txt
