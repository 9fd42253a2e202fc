#!/bin/bash
# round 3: instruction-cache / issue counters of two builds of the fused kernels, side by side (rocprofv3 --pmc over kbench)
#   tools/kbench/pmc_r3.sh <tag> <lib.so> [more "tag lib" pairs ...]
set +e
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r3/pmc
mkdir -p $OUT
while [ $# -ge 2 ]; do
  tag=$1; lib=$2; shift 2
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
    s=$(echo $set | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o $s -- tools/kbench/kbench $lib 512 512 256 4 "PP_FUSED_SUM=1" > $OUT/$tag.$s.log 2>&1
  done
  python tools/pmc_summary.py $OUT/$tag $OUT/$tag.md > /dev/null 2>&1
  echo "== $tag"; grep "k_fused2" $OUT/$tag.md
done
