#!/bin/bash
# compiler scheduling strategies on the fused kernels (pp_demons.hip rebuilt with -mllvm flags; tools/kbench/build_variants.sh)
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
V=tools/kbench/variants
MAIN=platipy_amd/csrc/libplatipy_hip.so
mkdir -p gpurun_out/r3
{
for rep in 1 2; do
  for lib in $MAIN $V/sched_ilp.so $V/sched_mem.so $V/sched_bias0.so $V/sched_bias100.so; do
    timeout 120 $KB $lib 512 512 256 30 "PP_FUSED_SUM=1"
  done
done
for lib in $MAIN $V/sched_ilp.so $V/sched_mem.so; do
  timeout 120 $KB $lib 341 341 171 40 "PP_FUSED_SUM=1"
done
} 2>&1 | tee gpurun_out/r3/kbench_r3_11.txt
