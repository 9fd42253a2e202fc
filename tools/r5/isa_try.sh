#!/bin/bash
# One line of static ISA figures per compile-flag set for the two generation-2 MASK kernels (CPU only; tools/kbench/mini.sh):
#   tools/r5/isa_try.sh "-fno-slp-vectorize" "-mllvm -slp-threshold=2" ...
cd "$(dirname "$0")/../.."
for flags in "$@"; do
  out=$(tools/kbench/mini.sh $flags 2>&1)
  va=$(echo "$out" | grep -E "VGPRs:|VGPRs Spill|SGPRs Spill" | tr '\n' ' ')
  for k in k_fused2_force_smooth k_fused2_add_smooth_warp; do
    python - "$k" "$flags" <<'PY'
import collections, re, sys
k, flags = sys.argv[1], sys.argv[2]
ops = collections.Counter()
for line in open(f"/tmp/mini_{k}.s"):
    m = re.match(r"\s+([a-z_0-9]+)\s", line)
    if m: ops[re.sub(r"_(e32|e64|dpp|sdwa)$", "", m.group(1))] += 1
tot = sum(ops.values())
valu = sum(v for o, v in ops.items() if o.startswith("v_"))
pk = sum(v for o, v in ops.items() if o.startswith("v_pk_"))
vmem = sum(v for o, v in ops.items() if o.startswith(("buffer_", "global_", "scratch_")))
lds = sum(v for o, v in ops.items() if o.startswith("ds_"))
salu = sum(v for o, v in ops.items() if o.startswith("s_") and o not in ("s_waitcnt", "s_nop", "s_barrier"))
print(f"{flags or '(default)':44s} {k[9:22]:14s} total {tot:5d} valu {valu:5d} (mov {ops['v_mov_b32'] + ops['v_mov_b64']:4d} pk {pk:4d} cnd {ops['v_cndmask_b32']:4d}) salu {salu:5d} lds {lds:4d} vmem {vmem:4d} wait {ops['s_waitcnt']:4d} nop {ops['s_nop']:4d} scratch {sum(v for o, v in ops.items() if o.startswith('scratch_'))}")
PY
  done
  echo "    $va"
done
