#!/bin/bash
# Kernel timelines of the atlas chains of one GPU under each schedule: tools/r6/timeline.sh [atlases] [streams]
# -> gpurun_out/r6_tl_<schedule>.md
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
A=${1:-4}; S=${2:-4}
for sched in "lockstep 0 0" "turnstile 1 0" "slots1 1 1"; do
  set -- $sched
  rm -rf /tmp/tl_$1
  TL_STAGGER=$2 TL_SLOTS=$3 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$1 -o tl -- python tools/streams_timeline.py run $A $S > gpurun_out/r6_tl_$1.log 2>&1
  python tools/streams_timeline.py analyse /tmp/tl_$1 "Round 6: $A atlas chains on $S HIP streams, schedule '$1' (STAGGER=$2, ENTRY_SLOTS=$3)" > gpurun_out/r6_tl_$1.md
  grep TIMELINE gpurun_out/r6_tl_$1.log >> gpurun_out/r6_tl_$1.md
done
