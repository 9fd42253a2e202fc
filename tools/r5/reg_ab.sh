#!/bin/bash
# round 5: config 2's whole registration, kernel by kernel, under an A/B environment switch
#   tools/r5/reg_ab.sh [VAR=value]      (default: PP_RESAMPLE_GENERIC=1 -- the general resample kernels against the axis-aligned ones)
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
SW=${1:-PP_RESAMPLE_GENERIC=1}
for mode in default switched; do
  mkdir -p gpurun_out/reg_$mode
  if [ $mode = switched ]; then export $SW; fi
  for i in 1 2 3; do python tools/profile_registration.py 2>/dev/null | grep registration_s; done
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/reg_$mode -o reg -- python tools/profile_registration.py > gpurun_out/reg_$mode/run.log 2>&1
  echo "== $mode ($([ $mode = switched ] && echo $SW || echo library defaults))"; grep registration_s gpurun_out/reg_$mode/run.log
  python tools/rocpd_stats.py gpurun_out/reg_$mode/reg_results.db | head -34 | cut -c1-150
done
