#!/bin/bash
# round 5: HBM-side bytes per voxel of every kernel of config 2's registration (two PMC passes, kernel-trace only)
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
OUT=gpurun_out/reg_pmc; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OLDPWD/$OUT -o $c -- bash -c "cd $OLDPWD && python tools/profile_registration.py" > $OLDPWD/$OUT/$c.log 2>&1 )
done
python tools/r5/reg_pmc.py $OUT
