#!/bin/bash
# round 3, trace 1: per-wave barrier timeline of the two fused kernels (PP_TRACE builds)
cd "$(dirname "$0")/../.."
KB=tools/kbench/kbench
V=tools/kbench/variants
mkdir -p gpurun_out/r3
{
timeout 120 $KB $V/trace_nofast.so 512 512 256 6 "PP_FUSED_SUM=1"
timeout 120 $KB $V/trace.so 512 512 256 6 "PP_FUSED_SUM=1"
} 2>&1 | tee gpurun_out/r3/kbench_r3_2.txt
