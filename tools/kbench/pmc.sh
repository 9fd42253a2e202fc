#!/bin/bash
# rocprofv3 PMC passes over the kbench harness (each counter set in its own run, kernel-trace only), then a
# per-kernel summary.  usage: tools/kbench/pmc.sh <tag> [lib.so]   -> gpurun_out/pmc_<tag>/summary.md
set +e
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
TAG=${1:-run}
LIB=${2:-platipy_amd/csrc/libplatipy_hip.so}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  KB_CALIBRATE=1 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o $tag -- tools/kbench/kbench $LIB 512 512 256 4 "PP_FUSED_GEN=2" > $OUT/$tag.log 2>&1
  tail -1 $OUT/$tag.log
done
python tools/pmc_summary.py $OUT $OUT/summary.md
cat $OUT/summary.md
