#!/usr/bin/env python3
"""Instruction histogram of one disassembled kernel (tools/kbench/mini.sh writes /tmp/mini_<kernel>.s).

    python tools/r5/isa_stats.py /tmp/mini_k_fused2_force_smooth.s [--top 25]

Classes: vector ALU (v_*), scalar (s_* except waits / nops / barriers), LDS (ds_*), vector memory (buffer_* / global_*),
waits.  A v_cndmask whose VCC / SGPR-pair mask was last written by a scalar instruction is counted separately
(profiles/round4_valu_rate_probe.txt prices it at ~22 cycles against ~3.5 behind a vector compare).
"""
import collections
import re
import sys


def parse(path):
    ins = []
    for line in open(path):
        m = re.match(r"\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if m:
            ins.append((m.group(1), m.group(2)))
    return ins


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    ins = parse(path)
    cls = collections.Counter()
    ops = collections.Counter()
    vcc_writer = None
    scalar_vcc_sel = 0
    for op, args in ins:
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        ops[base] += 1
        if op.startswith("v_"):
            cls["valu"] += 1
            if base.startswith("v_cmp") and (args.startswith("vcc") or op.endswith("e32")):
                vcc_writer = "v"
            if base == "v_cndmask_b32" and ("vcc" in args) and vcc_writer == "s":
                scalar_vcc_sel += 1
        elif op in ("s_waitcnt",):
            cls["waitcnt"] += 1
        elif op in ("s_nop",):
            cls["s_nop"] += 1
        elif op == "s_barrier":
            cls["barrier"] += 1
        elif op.startswith("s_"):
            cls["salu"] += 1
            if args.startswith("vcc"):
                vcc_writer = "s"
        elif op.startswith("ds_"):
            cls["lds"] += 1
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
            cls["vmem"] += 1
        else:
            cls["other"] += 1
    print(f"{path}: {len(ins)} instructions")
    print("  " + ", ".join(f"{k} {v}" for k, v in sorted(cls.items(), key=lambda kv: -kv[1])))
    print(f"  v_cndmask behind a scalar-written VCC: {scalar_vcc_sel}")
    for fam in ("v_mov_b32", "v_cndmask_b32", "v_pk_fma_f32", "v_fma_f32", "v_fmac_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mul_f32", "v_add_f32", "v_sub_f32"):
        print(f"  {fam}: {ops.get(fam, 0)}", end="")
    print()
    print("  top: " + ", ".join(f"{k} {v}" for k, v in ops.most_common(top)))


if __name__ == "__main__":
    main()
