#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (SQLite) capture: per-kernel calls / total / average / min / max
duration, the table `rocprofv3 --stats` prints.  Usage: rocpd_stats.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start), "
                     f"max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | VGPR | SGPR | LDS B |", file=out)
    print("|---|---|---|---|---|---|---|---|---|---|", file=out)
    for n, cnt, s, a, mn, mx, vg, sg, lds in rows:
        n = n if len(n) < 90 else n[:87] + "..."
        print(f"| `{n}` | {cnt} | {s/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*s/tot:.1f} | {vg} | {sg} | {lds} |", file=out)


if __name__ == "__main__":
    main()
