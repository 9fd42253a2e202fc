#!/bin/bash
set +e
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for opt in 4 2; do
  export PP_FUSED_OPT=$opt
  echo "#### OPT=$opt"; bash tools/gpu_quick.sh 2>&1 | grep -v amdgpu.ids
done
