#!/usr/bin/env python
"""Kernel durations and end->start gaps from rocprofv3 --kernel-trace CSVs (tools/kbench/run_r3_10.sh)."""
import collections
import csv
import glob
import re
import sys

for d in sys.argv[1:]:
    fs = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True))
    if not fs:
        print(d, "no trace")
        continue
    rows = list(csv.DictReader(open(fs[-1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = collections.defaultdict(list)
    gaps = []
    prev_end = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        m = re.search(r"(k_\w+|__amd\w+)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:30]
        dur[name].append(e - s)
        if prev_end is not None and ("force_smooth" in name or "add_smooth_warp" in name):
            gaps.append(s - prev_end)
        prev_end = e
    print(d)
    for k, v in dur.items():
        v2 = sorted(v)
        print(f"  {k:36s} n={len(v):5d} median {v2[len(v2) // 2] / 1000:8.2f} us  mean {sum(v) / len(v) / 1000:8.2f} us  min {v2[0] / 1000:.2f}")
    g = sorted(gaps)
    if g:
        print(f"  end->start gap before a fused kernel: median {g[len(g) // 2] / 1000:.2f} us  mean {sum(g) / len(g) / 1000:.2f} us  n={len(g)}")
