#!/bin/bash
# second set of rocprofv3 PMC passes over the kbench harness: instruction cache, issue cycles per pipe, TA / TCP stalls
set +e
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
OUT=gpurun_out/pmc_x
mkdir -p $OUT
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_THREAD_CYCLES_VALU" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  KB_CALIBRATE=0 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o $tag -- tools/kbench/kbench platipy_amd/csrc/libplatipy_hip.so 512 512 256 4 "PP_FUSED_SUM=1" > $OUT/$tag.log 2>&1
  tail -1 $OUT/$tag.log | cut -c1-120
done
python tools/pmc_summary.py $OUT $OUT/summary.md
grep "k_fused2" $OUT/summary.md
