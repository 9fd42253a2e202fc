#!/usr/bin/env python3
"""Reduce tools/r6/linear_pmc.sh's counter passes: per metric kernel, sampling mode and GRID SIZE (= pyramid level), mean counter
values per launch.  tools/profile_linear.py runs its ITK-sampling registrations first and the lattice ones afterwards: a launch
that starts before the pass's first k_metric_grad<0, false> belongs to the ITK runs."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]


def short(name):
    m = re.search(r"(k_metric\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else None


agg = defaultdict(lambda: defaultdict(list))     # (kernel, mode, grid) -> counter -> values
dur = defaultdict(list)
for tf in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(tf)) if short(r.get("Kernel_Name", ""))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cut = next((int(r["Start_Timestamp"]) for r in rows if "k_metric_grad<0, false>" in r["Kernel_Name"]), None)
    mode_of = {}
    for r in rows:
        mode = "ITK sampling" if (cut is None or int(r["Start_Timestamp"]) < cut) else "lattice"
        mode_of[r["Dispatch_Id"]] = mode
        g = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
        dur[(short(r["Kernel_Name"]), mode, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    cf = tf.replace("kernel_trace", "counter_collection")
    if os.path.exists(cf):
        for r in csv.DictReader(open(cf)):
            k = short(r.get("Kernel_Name", ""))
            if k:
                agg[(k, mode_of.get(r["Dispatch_Id"], "?"), int(r.get("Grid_Size") or 0))][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for v in agg.values() for c in v})
print("| kernel | sampling | grid (threads) | launches | avg us | " + " | ".join(counters) + " |")
print("|---|---|---|---|---|" + "---|" * len(counters))
for key in sorted(agg, key=lambda k: (k[0], k[2], k[1])):
    v = agg[key]
    us = dur.get(key, [])
    print(f"| {key[0]} | {key[1]} | {key[2]} | {len(us) // max(1, len(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)))} | "
          f"{sum(us) / len(us) if us else float('nan'):.1f} | " + " | ".join(f"{sum(v[c]) / len(v[c]):.3g}" if c in v else "-" for c in counters) + " |")
