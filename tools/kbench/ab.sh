#!/bin/bash
# A/B timing of builds of libplatipy_hip.so on the fused demons iteration (tools/kbench/kbench), alternating over the builds so
# that box drift (the clocks sag as the package heats up) hits all of them alike.
#   tools/kbench/ab.sh [-s "NX NY NZ"] [-n iterations] [-r rounds] [-p "sx,sy,sz"] build[:ENV=V,ENV2=V ...] ...
# build: "main" (platipy_amd/csrc/libplatipy_hip.so), a name under tools/kbench/variants/ (tools/kbench/build_variants.sh name
# "-DFLAGS" ...), or a path to a .so.  Each ":"-suffix is one env spec for that build (kbench applies and clears it per run).
#   tools/kbench/ab.sh -r 3 r4 main noflip:PP_FUSED_GEN=2
#   tools/kbench/ab.sh -s "341 341 171" -p 1.5,1.5,1.5 main syncaw:PP_FUSED_SYNC=0:PP_FUSED_SYNC=4
cd "$(dirname "$0")/../.."
SIZE="512 512 256"; ITERS=60; ROUNDS=2; SPACING=""
while getopts "s:n:r:p:" o; do
  case $o in s) SIZE=$OPTARG;; n) ITERS=$OPTARG;; r) ROUNDS=$OPTARG;; p) SPACING=$OPTARG;; *) exit 2;; esac
done
shift $((OPTIND - 1))
[ -n "$SPACING" ] && export KB_SPACING=$SPACING
for rep in $(seq 1 $ROUNDS); do
  for spec in "$@"; do
    name=${spec%%:*}
    case $name in main) lib=platipy_amd/csrc/libplatipy_hip.so;; */*|*.so) lib=$name;; *) lib=tools/kbench/variants/$name.so;; esac
    [ -f "$lib" ] || { echo "no such build: $lib"; continue; }
    envs=()
    if [ "$spec" != "$name" ]; then IFS=':' read -ra envs <<< "${spec#*:}"; else envs=("PP_FUSED_GEN=2"); fi
    timeout 300 tools/kbench/kbench $lib $SIZE $ITERS "${envs[@]}" | cut -c1-230
  done
done
