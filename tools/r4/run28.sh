#!/bin/bash
# round 4, GPU visit 28: the evidence files of the round's second half (linear-stage kernels, level-size demons launches, the chain)
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
O=gpurun_out/r4b
mkdir -p $O
{
  echo "== round-3 metric kernels (PP_METRIC_LANES=0 PP_METRIC_GRAD_ONE_LAUNCH=0)"
  PP_METRIC_LANES=0 PP_METRIC_GRAD_ONE_LAUNCH=0 timeout 200 python tools/profile_linear.py
  echo "== round-4 metric kernels (defaults)"
  timeout 200 python tools/profile_linear.py
  echo "== determinism under alternating inputs"
  timeout 300 python tools/r4/metric_stress.py 2000
} > $O/linear_stage.txt 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OLDPWD/$O/lin -o lin -- bash -c "cd $OLDPWD && python tools/profile_linear.py" > $OLDPWD/$O/lin_run.log 2>&1 )
python tools/rocpd_by_grid.py $O/lin/lin_results.db metric > $O/linear_kernels_by_level.txt 2>&1
rm -rf $O/lin
KB=tools/kbench/kbench
MAIN=platipy_amd/csrc/libplatipy_hip.so
{
  export KB_SPACING=1.5,1.5,1.5
  for size in "341 341 171 30" "340 340 170 30" "405 405 200 20" "171 171 85 60" "85 85 43 100"; do
    echo "== $size  (PITCH: padded rows, MIX: mixed tile shapes; 1 forces, 0 forbids, unset: the launcher's rule)"
    timeout 120 $KB $MAIN $size "PP_FUSED_GEN=2" "PP_FUSED_PITCH=0,PP_FUSED_MIX=0" "PP_FUSED_PITCH=1,PP_FUSED_MIX=0" "PP_FUSED_PITCH=0,PP_FUSED_MIX=1" "PP_FUSED_PITCH=1,PP_FUSED_MIX=1"
  done
} > $O/kbench_level_sizes.txt 2>&1
{
  timeout 300 python tools/r4/chain_stages.py
  timeout 300 python tools/r4/chain_levels.py
} > $O/chain_stages.txt 2>&1
tail -5 $O/linear_stage.txt; cat $O/linear_kernels_by_level.txt; cut -c1-200 $O/kbench_level_sizes.txt; tail -12 $O/chain_stages.txt
