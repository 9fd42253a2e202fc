#!/usr/bin/env python
"""HBM-side bytes of the once-per-level kernels of config 2's registration: reduces the two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; tools/r5/reg_pmc.sh) to bytes per voxel of the largest dispatch of each kernel.
FETCH_SIZE / WRITE_SIZE are KiB; the factors (gfx950: fetch x 2.000, write x 1.000) are the ones calibrated in the same
build's capture, profiles/round5_pmc.json (tools/pmc_reduce.py)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
cal = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "profiles", "round5_pmc.json")))
ff, wf = cal["fetch_factor"], cal["write_factor"]
N = 512 * 512 * 256
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(?:<[^>]*>)?)", row["Kernel_Name"])
        if not m:
            continue
        acc[m.group(1)][row["Counter_Name"]].append((float(row["Counter_Value"]), int(row["Grid_Size"])))
print("| kernel (largest dispatch) | dispatches | fetched B/voxel | written B/voxel | sum |")
print("|---|---|---|---|---|")
rows = []
for k, c in acc.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    gmax = max(g for _, g in c["FETCH_SIZE"])
    fe = [v for v, g in c["FETCH_SIZE"] if g == gmax]
    wr = [v for v, g in c["WRITE_SIZE"] if g == gmax]
    fb = sum(fe) / len(fe) * 1024.0 * ff / N
    wb = sum(wr) / len(wr) * 1024.0 * wf / N
    rows.append((fb + wb, k, len(fe), fb, wb))
for s, k, n, fb, wb in sorted(rows, reverse=True):
    if s > 0.5:
        print(f"| {k} | {n} | {fb:.2f} | {wb:.2f} | {s:.2f} |")
