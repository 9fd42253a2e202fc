#!/bin/bash
# round 5: energy budget of the fused demons iteration.  For the product build and each ablation
# variant of pp_demons.hip: package power (rocm-smi) over ~3000 back-to-back iterations, per-kernel times, and the effective
# shader clock GRBM_GUI_ACTIVE / (8 XCDs x dispatch duration) from a separate --pmc pass (MI355X_MICROARCH.md's method).
#   tools/r5/energy_budget.sh [out dir]      -> <out>/energy_budget.txt (tools/r5/energy_table.py turns it into the table)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
KB=tools/kbench/kbench
V=tools/kbench/variants
OUT=${1:-gpurun_out/r5/energy}
mkdir -p $OUT
{
rocm-smi --showmaxpower 2>&1 | grep -i "max"
for tag in main ablA_nomem ablA_noload ablA_nostore ablB_nomem abl_noload abl_nostore abl_nogather; do
  case $tag in main) lib=platipy_amd/csrc/libplatipy_hip.so;; *) lib=$V/$tag.so;; esac
  [ -f $lib ] || continue
  echo "== $tag"
  tools/r5/power_sample.sh $tag timeout 120 $KB $lib 512 512 256 1500 "PP_FUSED_GEN=2" | cut -c1-200 | sed "s/^/RUN $tag: /" | sed "s/^RUN $tag: POWER/POWER/"
  rm -rf $OUT/clk_$tag
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/clk_$tag -o clk -- $KB $lib 512 512 256 6 "PP_FUSED_GEN=2" > $OUT/clk_$tag.log 2>&1
  python tools/r5/clk_from_pmc.py $OUT/clk_$tag $tag
done
echo "== idle"
sleep 2; for i in 1 2 3; do rocm-smi --showpower 2>/dev/null | grep -i "power (W)" | head -1; done
} 2>&1 | tee $OUT/energy_budget.txt
find $OUT -name "*.csv" -size +2M -delete
