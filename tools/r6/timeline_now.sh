cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
rm -rf /tmp/tl_now
TL_STAGGER=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_now -o tl -- python tools/streams_timeline.py run 4 4 > gpurun_out/r6_tl_now.log 2>&1
python tools/streams_timeline.py analyse /tmp/tl_now "Round 6 (final build, ITK sampling default): 4 atlas chains on 4 HIP streams, lockstep" > gpurun_out/r6_tl_now.md
grep TIMELINE gpurun_out/r6_tl_now.log >> gpurun_out/r6_tl_now.md
