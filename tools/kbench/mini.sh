#!/bin/bash
# Fast compile of pp_demons.hip with only the radius-2 / 64x16 / SUM / nt instances (PP_MINI) for ISA inspection:
#   tools/kbench/mini.sh [extra -D flags]   ->  /tmp/mini.s (disassembly), resource usage of the two generation-2 kernels on stdout
cd "$(dirname "$0")/../.."
set -e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-function -DPP_MINI "$@" \
  -Rpass-analysis=kernel-resource-usage -c platipy_amd/csrc/pp_demons.hip -o /tmp/mini.o 2> /tmp/mini.log || { tail -30 /tmp/mini.log; exit 1; }
grep -A12 "Function Name: _ZN12_GLOBAL__N_1.*k_fused2.*ILi2ELi0ELb1ELb1ELb1E" /tmp/mini.log | grep -E "Function Name|VGPRs:|TotalSGPRs|Occupancy|ScratchSize|LDS Size|Spill" | sed 's/.*remark: *//;s/\[-Rpass.*//'
(cd /tmp && rm -f mini.o.0.* && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading mini.o > /dev/null 2>&1; /opt/rocm/lib/llvm/bin/llvm-objdump -d mini.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 > mini.s)
for k in k_fused2_force_smooth k_fused2_add_smooth_warp; do
  awk -v k="$k" '$0 ~ "^[0-9a-f]+ <.*"k"ILi2ELi0ELb1ELb1ELb1E" {on=1; print; next} on && /^[0-9a-f]+ </ {exit} on {print}' /tmp/mini.s > /tmp/mini_$k.s
  echo "$k: $(wc -l < /tmp/mini_$k.s) lines, s_nop $(grep -c s_nop /tmp/mini_$k.s), v_readlane $(grep -c v_readlane /tmp/mini_$k.s), vmcnt(0) $(grep -c 'vmcnt(0)' /tmp/mini_$k.s)"
done
