#!/bin/bash
# rebuild libplatipy_hip.so if stale, then run the given command on the GPU box:  tools/g.sh [--timeout S] 'cmd'
cd "$(dirname "$0")/.."
python -c "from platipy_amd._build import build_hip; build_hip()" || exit 1
T=1500
if [ "$1" == "--timeout" ]; then T=$2; shift 2; fi
gpurun --timeout $T -- "$@" 2>&1 | grep -v "^\[gpurun\] send\|amdgpu.ids"
