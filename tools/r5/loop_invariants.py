#!/usr/bin/env python3
"""Vector registers a kernel's plane loop reads but never writes (loop-invariant per-thread state), from a disassembly.
    python tools/r5/loop_invariants.py /tmp/mini_k_fused2_force_smooth.s FIRST_LINE LAST_LINE
"""
import re
import sys

path, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lines = open(path).read().split("\n")[lo - 1:hi]


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


written, read = set(), {}
for ln in lines:
    m = re.match(r"\s+([a-z_0-9]+)\s+(.*?)\s*//", ln)
    if not m:
        continue
    op, args = m.group(1), m.group(2)
    toks = [t.strip() for t in re.split(r",(?![^\[]*\])", args)]
    stores = op.startswith(("ds_write", "buffer_store", "global_store", "scratch_store", "v_cmp", "v_cmpx"))
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        stores = True
    dst = [] if stores or not toks else regs(toks[0])
    src_toks = toks if stores else toks[1:]
    for t in src_toks:
        for r in regs(t):
            read.setdefault(r, []).append(op)
    if op.startswith("v_fmac") or op.startswith("v_mac"):   # dst is also a source
        for r in dst:
            read.setdefault(r, []).append(op)
    written.update(dst)
inv = sorted(r for r in read if r not in written)
print(f"{len(inv)} loop-invariant VGPRs: {inv}")
for r in inv:
    ops = read[r]
    print(f"  v{r}: {len(ops)} reads, e.g. {sorted(set(ops))[:6]}")
