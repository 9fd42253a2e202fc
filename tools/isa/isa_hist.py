#!/usr/bin/env python
"""Instruction-class histogram of one kernel in a hipcc -S listing (tools/isa: ISA inspection helpers).
Usage: isa_hist.py file.s <substring of the mangled kernel name> [--blocks]"""
import collections
import re
import sys


def kernel_lines(path, key):
    out, on = [], False
    for ln in open(path):
        if not on:
            if ln.startswith("_ZN") and key in ln and re.match(r"^_ZN\w+:", ln):
                on = True
            continue
        if ln.startswith(".Lfunc_end"):
            break
        out.append(ln.rstrip("\n"))
    return out


def klass(m):
    if m.startswith("s_waitcnt"):
        return "s_waitcnt"
    if m.startswith("s_barrier"):
        return "s_barrier"
    if m.startswith("s_cbranch") or m.startswith("s_branch"):
        return "s_branch"
    if m.startswith("s_load") or m.startswith("s_buffer"):
        return "smem"
    if m.startswith("s_"):
        return "salu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith("buffer_") or m.startswith("global_") or m.startswith("flat_") or m.startswith("scratch_"):
        return "vmem"
    if m.startswith("v_readlane") or m.startswith("v_writelane") or m.startswith("v_readfirstlane"):
        return "v_lane"
    if m.startswith("v_"):
        return "valu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = kernel_lines(path, key)
    blocks, cur = [], ["entry", []]
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", s):
                blocks.append(cur)
                cur = [s.split(":")[0], []]
            continue
        m = s.split()[0]
        cur[1].append(m)
    blocks.append(cur)
    tot = collections.Counter()
    for name, ins in blocks:
        c = collections.Counter(klass(m) for m in ins)
        tot.update(c)
        if "--blocks" in sys.argv and len(ins) >= 40:
            top = collections.Counter(ins).most_common(8)
            print(f"{name:12s} n={len(ins):5d} ", dict(c), top)
    print("TOTAL", sum(tot.values()), dict(tot))
    allm = collections.Counter(m for _, ins in blocks for m in ins)
    print(allm.most_common(45))


if __name__ == "__main__":
    main()
