#!/bin/bash
# round 5: config 2's whole registration, kernel by kernel, marching gathers against the general ones
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
for mode in march generic; do
  mkdir -p gpurun_out/reg_$mode
  if [ $mode = generic ]; then export PP_RESAMPLE_GENERIC=1; fi
  for i in 1 2 3; do python tools/profile_registration.py | grep registration_s; done
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/reg_$mode -o reg -- python tools/profile_registration.py > gpurun_out/reg_$mode/run.log 2>&1
  echo "== $mode"; grep registration_s gpurun_out/reg_$mode/run.log
  python tools/rocpd_stats.py gpurun_out/reg_$mode/reg_results.db | head -34 | cut -c1-150
done
