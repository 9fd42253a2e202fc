#!/bin/bash
# Generation 3 (pp_demons_cube.h, bricks with their halo in LDS) against the marching kernels on the latency-bound level sizes:
# tools/kbench per size, 200 iterations, PP_FUSED_CUBE=0 / 1 alternating twice; equal checksums = bit-identical fields.
#   bash tools/r6/cube_levels.sh > gpurun_out/cube_levels.txt
cd "$(dirname "$0")/../.."
LIB=platipy_amd/csrc/libplatipy_hip.so
[ -x tools/kbench/kbench ] || hipcc -O2 --offload-arch=gfx950 -o tools/kbench/kbench tools/kbench/kbench.cpp -ldl
for s in "43 40 28" "64 64 32" "72 72 50" "87 80 57" "96 96 48" "80 80 80" "100 100 60" "100 100 100" "128 128 64" "160 160 96"; do
  echo "== $s"
  tools/kbench/kbench $LIB $s 200 "PP_FUSED_CUBE=0" "PP_FUSED_CUBE=1" "PP_FUSED_CUBE=0" "PP_FUSED_CUBE=1" | cut -c1-260
done
