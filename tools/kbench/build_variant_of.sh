#!/bin/bash
# A/B variants of libplatipy_hip.so that differ in how ONE source file is compiled (measurement tooling):
#   tools/kbench/build_variant_of.sh pp_fusion name1 "-DFLAG" name2 "..."   -> tools/kbench/variants/<name>.so
set -e
cd "$(dirname "$0")/../.."
python -c "from platipy_amd._build import build_hip; build_hip()"
mkdir -p tools/kbench/variants
CS=platipy_amd/csrc
SRC=$1; shift
OTHERS=$(ls $CS/*.o | grep -v "$SRC.o")
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-function $flags -c $CS/$SRC.hip -o tools/kbench/variants/$name.o 2>&1 | grep -v "hip-link\|Wpass-failed\|^\s*[0-9]* |\|__launch_bounds__\|In file included\|warning generated" || true
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/kbench/variants/$name.so tools/kbench/variants/$name.o $OTHERS
    rm -f tools/kbench/variants/$name.o
    echo "built $name ($flags)"
  ) &
done
wait
