#!/usr/bin/env python
"""Per (kernel, grid size) duration statistics of a rocprofv3 rocpd capture: which LEVEL of a pyramid a launch belongs to
shows in its grid.  Usage: rocpd_by_grid.py results.db [name substring]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
if gx is None:
    print("columns:", cols)
    sys.exit(1)
gy = gx.replace("_x", "_y")
like = f"%{sys.argv[2]}%" if len(sys.argv) > 2 else "%"
rows = c.execute(f"select {name_col}, {gx}, {gy}, count(*), avg(end - start), min(end - start), max(end - start), sum(end - start) from kernels "
                 f"where {name_col} like ? group by {name_col}, {gx}, {gy} order by 8 desc", (like,)).fetchall()
for n, x, y, cnt, a, mn, mx, s in rows:
    print(f"{n[:70]:70s} grid {x}x{y} calls {cnt:5d} avg {a/1e3:8.1f} us min {mn/1e3:8.1f} max {mx/1e3:8.1f} total {s/1e6:8.3f} ms")
