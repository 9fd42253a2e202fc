#!/bin/bash
# Build A/B variants of libplatipy_hip.so that differ only in how pp_demons.hip is compiled (measurement tooling).
#   tools/kbench/build_variants.sh name1 "-DFLAG1 -DFLAG2" name2 "..." ...
# -> tools/kbench/variants/<name>.so ; every other object is the product build's (platipy_amd/csrc/*.o).
set -e
cd "$(dirname "$0")/../.."
python -c "from platipy_amd._build import build_hip; build_hip()"
mkdir -p tools/kbench/variants
CS=platipy_amd/csrc
OTHERS=$(ls $CS/*.o | grep -v pp_demons.o)
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-function $flags -c $CS/pp_demons.hip -o tools/kbench/variants/$name.o 2>&1 | grep -v "hip-link\|Wpass-failed\|^\s*[0-9]* |\|__launch_bounds__\|In file included\|warning generated" || true
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/kbench/variants/$name.so tools/kbench/variants/$name.o $OTHERS
    rm -f tools/kbench/variants/$name.o
    echo "built $name ($flags)"
  ) &
done
wait
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o tools/kbench/kbench tools/kbench/kbench.cpp -ldl 2>&1 | grep -v hip-link || true
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o tools/kbench/sbench tools/kbench/sbench.cpp -ldl 2>&1 | grep -v hip-link || true
ls -la tools/kbench/variants
