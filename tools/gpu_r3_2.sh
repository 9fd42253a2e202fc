#!/bin/bash
# ring FIR passes: GPU parity tests, stand-alone rates, the registration's per-kernel breakdown
set +e
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3b
mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels.py tests/test_registration.py -m gpu -q -k "fir or sparse or smooth_and_resample or discrete" 2>&1 | tail -5 | tee $OUT/pytest.log
timeout 300 tools/kbench/sbench platipy_amd/csrc/libplatipy_hip.so 512 512 256 5 2>&1 | tee $OUT/sbench.txt
PP_FIR_MARCH_SP=0 timeout 300 tools/kbench/sbench platipy_amd/csrc/libplatipy_hip.so 512 512 256 5 2>&1 | grep -i "gauss" | sed 's/^/PP_FIR_MARCH_SP=0 /' | tee -a $OUT/sbench.txt
bash tools/gpu_reg_prof.sh 2>&1 | head -40 | cut -c1-150 | tee $OUT/reg.txt
