#!/usr/bin/env python3
"""Effective shader clock of the fused demons kernels from a `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` pass:
clock = GRBM_GUI_ACTIVE / (XCDs x dispatch duration) (MI355X_MICROARCH.md's method; the counter is summed over the 8 XCDs of
the device -- 8.6e6 cycles in 546 us would be 15.8 GHz otherwise), mean over the launches of each kernel.

    python tools/r5/clk_from_pmc.py <output dir> <label>
"""
import csv
import glob
import os
import sys
from collections import defaultdict

d, label = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r.get("Dispatch_Id") or r.get("Correlation_Id")] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        name = r["Kernel_Name"]
        key = "A k_fused2_force_smooth" if "k_fused2_force_smooth" in name else "B k_fused2_add_smooth_warp" if "k_fused2_add_smooth_warp" in name else None
        if key is None:
            continue
        ns = None
        if r.get("End_Timestamp") and r.get("Start_Timestamp"):
            ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if not ns:
            ns = dur.get(r.get("Dispatch_Id"))
        if ns:
            acc[key].append((float(r["Counter_Value"]), ns, int(r["Grid_Size"])))
for key in sorted(acc):
    gmax = max(g for _, _, g in acc[key])
    rows = [(c, ns) for c, ns, g in acc[key] if g == gmax]
    cyc = sum(c for c, _ in rows) / len(rows)
    ns = sum(n for _, n in rows) / len(rows)
    print(f"CLOCK {label}: {key}: {len(rows)} launches, GRBM_GUI_ACTIVE {cyc:.4g} cycles (8 XCDs) in {ns / 1e3:.1f} us -> {cyc / ns * 1e3 / 8:.0f} MHz per XCD")
if not acc:
    print(f"CLOCK {label}: no GRBM_GUI_ACTIVE rows under {d}")
