#!/usr/bin/env python
"""Reduce a pmc_summary.py table to calibrated HBM-side bytes per launch (-> profiles/roundN_pmc.json, read by bench.py).

usage: pmc_reduce.py <pmc_counters.md> <commit> nx ny nz
Calibration (MI355X_MICROARCH.md, HBM section): kernels of the same run with known byte counts -- torch's elementwise
add and the library's fuse_divide (16 B/lane: read 12 N resp. 24 N bytes, write 12 N) and the sum-of-squared-differences
reduction (4 B/lane loads: reads 24 N bytes, writes nothing) -- give the factor that turns FETCH_SIZE / WRITE_SIZE (KiB)
into bytes for each access width; the fused kernels (16-B strips, 4-B and 8-B gathers) use the larger read factor, an
upper bound when the two differ (they agree: 2.0)."""
import json
import re
import sys

md, commit, nx, ny, nz = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
n = nx * ny * nz
tab = {}
for line in open(md):
    m = re.match(r"\| (\S.*?) \| (\S+) \| (\d+) \| (\S+) \|", line)
    if m and m.group(2) != "counter":
        tab.setdefault(m.group(1), {})[m.group(2)] = float(m.group(4))
cal = {}
if "torch.add(scalar)" in tab and "FETCH_SIZE" in tab["torch.add(scalar)"]:
    cal["fetch_16B"] = 12.0 * n / (tab["torch.add(scalar)"]["FETCH_SIZE"] * 1024)
    cal["write_16B"] = 12.0 * n / (tab["torch.add(scalar)"]["WRITE_SIZE"] * 1024)
if "k_map4<op_fuse_divide>" in tab and "FETCH_SIZE" in tab["k_map4<op_fuse_divide>"]:
    cal["fetch_16B_lib"] = 24.0 * n / (tab["k_map4<op_fuse_divide>"]["FETCH_SIZE"] * 1024)
    cal["write_16B_lib"] = 12.0 * n / (tab["k_map4<op_fuse_divide>"]["WRITE_SIZE"] * 1024)
if "k_ssd_partial" in tab and "FETCH_SIZE" in tab["k_ssd_partial"]:
    cal["fetch_4B"] = 24.0 * n / (tab["k_ssd_partial"]["FETCH_SIZE"] * 1024)
ff = max(cal.get("fetch_16B", 2.0), cal.get("fetch_4B", 2.0), cal.get("fetch_16B_lib", 2.0))
fw = max(cal.get("write_16B", 1.0), cal.get("write_16B_lib", 1.0))
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha16  # noqa: E402  (the capture is tied to the kernel sources it was taken of)

raw = md.replace("gpurun_out/r2/", "profiles/round2_").replace("gpurun_out/r3final/", "profiles/round3_").replace("gpurun_out/r4final/", "profiles/round4_")
out = {"commit": commit, "kernel_source_sha16": kernel_source_sha16(), "size": [nx, ny, nz], "raw": raw, "calibration": cal,
       "fetch_factor": ff, "write_factor": fw, "hbm_bytes_per_launch": {}, "fetch_bytes_per_launch": {}, "write_bytes_per_launch": {},
       "tcc_hit_rate": {}}
for k, c in tab.items():
    if not k.startswith("k_fused") and not k.startswith("k_warp") and not k.startswith("k_demons_force") and not k.startswith("k_conv"):
        continue
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["fetch_bytes_per_launch"][k] = c["FETCH_SIZE"] * 1024 * ff
        out["write_bytes_per_launch"][k] = c["WRITE_SIZE"] * 1024 * fw
        out["hbm_bytes_per_launch"][k] = out["fetch_bytes_per_launch"][k] + out["write_bytes_per_launch"][k]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        out["tcc_hit_rate"][k] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
print(json.dumps(out, indent=1))
