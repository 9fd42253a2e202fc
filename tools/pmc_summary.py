#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSVs (one *_counter_collection.csv per pass) per kernel and counter:
mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    if "op_fuse_divide" in name:
        return "k_map4<op_fuse_divide>"
    for key in ("k_ssd_partial", "k_fused2_add_smooth_warp", "k_fused2_force_smooth", "k_cal_copy16", "k_cal_copy4", "k_fused_add_smooth_warp", "k_fused_force_smooth", "k_fuse_divide", "k_warp_same_grid", "k_demons_force",
                "k_conv_axis", "k_demons_finalize", "k_copy_if_odd"):
        if key in name:
            return key
    if "CUDAFunctorOnSelf_add" in name or "AUnaryFunctor" in name and "add" in name:
        return "torch.add(scalar)"
    return None


def main():
    d = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k is None:
                continue
            # keep the big (bench-size) dispatches only
            acc[k][row["Counter_Name"]].append((float(row["Counter_Value"]), int(row["Grid_Size"])))
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    print("| kernel | counter | dispatches | mean per dispatch |", file=out)
    print("|---|---|---|---|", file=out)
    for k in sorted(acc):
        for cname in sorted(acc[k]):
            vals = acc[k][cname]
            gmax = max(g for _, g in vals)
            big = [v for v, g in vals if g == gmax]
            print(f"| {k} | {cname} | {len(big)} | {sum(big)/len(big):.6g} |", file=out)


if __name__ == "__main__":
    main()
