#!/bin/bash
# rocprofv3 PMC passes (each counter set in its own run, kernel-trace only) for the fused demons kernels.
set +e
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/pmc
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc -o $tag -- python tools/pmc_probe.py > gpurun_out/pmc/$tag.log 2>&1
  tail -1 gpurun_out/pmc/$tag.log
done
ls -R gpurun_out/pmc | head -40
