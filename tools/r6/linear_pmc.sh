#!/bin/bash
# Counters of the linear stage's metric kernels (tools/profile_linear.py: quick similarity + the pipelines' affine registration at
# 512x512x256, ITK sampling on and off), one rocprofv3 --pmc pass per counter set (kernel-trace only) -> gpurun_out/r6_linpmc/
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6_linpmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L > $OUT/counters_available.txt 2>&1
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" \
           "SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OLDPWD/$OUT -o $tag -- bash -c "cd $OLDPWD && python tools/profile_linear.py" > $OLDPWD/$OUT/$tag.log 2>&1 )
  tail -c 300 $OUT/$tag.log | tr '\n' ' ' | cut -c1-300; echo
done
python tools/r6/linear_pmc.py $OUT
