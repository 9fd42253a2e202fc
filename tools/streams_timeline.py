#!/usr/bin/env python3
"""Which kernels of which HIP stream run when, for config 5's per-GPU shape (4 atlas chains on 4 streams).

    rocprofv3 --kernel-trace --output-format csv -d OUT -o tl -- python tools/streams_timeline.py run [atlases] [streams]
    python tools/streams_timeline.py analyse OUT "title" > profiles/round6_streams_timeline.md

`run` takes the schedule from the environment: TL_STAGGER=0|1 (projects.multiatlas.STAGGER), TL_SLOTS=n (ENTRY_SLOTS).

`run`: one warm-up pass of bench.multi_atlas_streams_leg's workload, 0.4 s of idle, then the pass that is analysed.
`analyse`: the dispatches after the last idle gap >= 0.25 s, per queue (= stream): busy time, kernel classes, and a timeline in
5 ms bins (one letter per queue and bin: the class that held most of the bin; lower case when the queue was busy < 50 % of it).
"""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(per_gpu, streams):
    sys.path.insert(0, ROOT)
    import time

    import torch

    import bench
    from platipy_amd import _lib

    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    real = bench.time.perf_counter
    state = {"n": 0}

    def marked():   # (the leg calls perf_counter exactly twice: before and after the timed pass)
        state["n"] += 1
        if state["n"] == 1:
            torch.cuda.synchronize()
            time.sleep(0.4)
        return real()

    from platipy_amd.projects import multiatlas

    if "TL_STAGGER" in os.environ:
        multiatlas.STAGGER = os.environ["TL_STAGGER"] != "0"
    if "TL_SLOTS" in os.environ:
        multiatlas.ENTRY_SLOTS = int(os.environ["TL_SLOTS"])
    print(f"TIMELINE_SCHEDULE stagger {multiatlas.STAGGER} entry slots {multiatlas.ENTRY_SLOTS}")
    bench.time.perf_counter = marked
    dt, dice, _ = bench.multi_atlas_streams_leg(ctx, (256, 512, 512), (1.0, 1.0, 1.0), dev, 0, 1, per_gpu=per_gpu, streams=streams)
    bench.time.perf_counter = real
    print(f"TIMELINE_RUN {per_gpu} atlases on {streams} streams: {dt:.4f} s (wall), dice {dice:.4f}")


def classify(name, threads):
    if "k_fused" in name:
        return "F" if threads >= 400 * 512 else "C"      # finest demons level (>= 400 blocks of 512) / coarse levels
    if "k_metric" in name or "k_mi_" in name or "k_meansq" in name or "k_corr" in name:
        return "L"                                      # linear registration's metric kernels
    if "k_rg_" in name or "k_compose" in name or "k_resample" in name or "k_warp" in name:
        return "R"                                      # resample / compose / recursive Gaussian between levels
    if "k_fir" in name or "k_gauss3" in name or "k_conv" in name:
        return "G"                                      # Gaussian blurs (pyramid, weight maps, fusion)
    return "o"


NAMES = {"F": "demons levels of >= 400 blocks", "C": "demons levels below that", "L": "linear metric", "R": "resample/compose/IIR", "G": "FIR blurs", "o": "other"}


def analyse(d, title="kernel timeline of the atlas chains of one GPU (512x512x256, pipeline defaults)"):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            gx = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id") or r.get("Stream_Id") or "?", r["Kernel_Name"], gx))
    rows.sort()
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - max(r[1] for r in rows[max(0, i - 50):i]) >= 250_000_000:
            cut = i
    rows = rows[cut:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    span = (t1 - t0) / 1e6
    print(f"# {title}\n")
    print(f"{len(rows)} dispatches in a device span of {span:.1f} ms (rocprofv3 --kernel-trace; dispatches after the pause before the timed pass).\n")
    queues = sorted({r[2] for r in rows})
    per_q = defaultdict(lambda: defaultdict(float))
    for s, e, q, name, gx in rows:
        per_q[q][classify(name, gx)] += (e - s) / 1e6
    print("| queue | dispatches | busy ms | " + " | ".join(NAMES[c] for c in "FCLRGo") + " |")
    print("|---|---|---|" + "---|" * 6)
    for q in queues:
        n = sum(1 for r in rows if r[2] == q)
        print(f"| {q} | {n} | {sum(per_q[q].values()):.1f} | " + " | ".join(f"{per_q[q][c]:.1f}" for c in "FCLRGo") + " |")
    tot = sum(sum(v.values()) for v in per_q.values())
    # union of busy intervals, and time with >= 2 queues busy
    ev = []
    for s, e, q, _, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, any_ms, multi_ms = 0, t0, 0.0, 0.0
    for t, dlt in ev:
        if depth >= 1:
            any_ms += (t - last) / 1e6
        if depth >= 2:
            multi_ms += (t - last) / 1e6
        depth += dlt
        last = t
    # the finest demons level: when it ran (union over queues), and the aggregate rate its voxel-iterations went at
    fin = sorted((s, e) for s, e, q, name, gx in rows if classify(name, gx) == "F")
    if fin:
        n_launch = len(fin)
        merged, (cs, ce) = [], fin[0]
        for s, e in fin[1:]:
            if s <= ce:
                ce = max(ce, e)
            else:
                merged.append((cs, ce))
                cs, ce = s, e
        merged.append((cs, ce))
        f_union = sum(e - s for s, e in merged) / 1e6
        f_multi = 0.0
        evf = sorted([(s, 1) for s, e in fin] + [(e, -1) for s, e in fin])
        depth, last = 0, evf[0][0]
        for t, dlt in evf:
            if depth >= 2:
                f_multi += (t - last) / 1e6
            depth += dlt
            last = t
        vox = float(os.environ.get("TL_FINEST_VOXELS", 341 * 341 * 171))
        print(f"\nDemons levels of >= 400 blocks: {n_launch} launches; one of them was running for {f_union:.1f} ms (two or more at once for "
              f"{f_multi:.1f} ms).")
    print(f"\nSum of kernel durations {tot:.1f} ms; device busy (>= 1 kernel running) {any_ms:.1f} ms of {span:.1f}; >= 2 kernels running {multi_ms:.1f} ms; idle {span - any_ms:.1f} ms.\n")
    binw = 5.0
    nb = int(span / binw) + 1
    print(f"Timeline, {binw:.0f} ms per column (F finest demons, C coarse demons, L linear metric, R resample/compose/IIR, G blurs, o other; lower case: queue busy < 50 % of the bin; . idle):\n")
    print("```")
    for q in queues:
        occ = [defaultdict(float) for _ in range(nb)]
        for s, e, qq, name, gx in rows:
            if qq != q:
                continue
            c = classify(name, gx)
            a, b = (s - t0) / 1e6, (e - t0) / 1e6
            i = int(a / binw)
            while a < b and i < nb:
                hi = min(b, (i + 1) * binw)
                occ[i][c] += hi - a
                a = hi
                i += 1
        line = ""
        for o in occ:
            busy = sum(o.values())
            if busy < 0.02 * binw:
                line += "."
            else:
                c = max(o, key=o.get)
                line += c if busy >= 0.5 * binw else c.lower()
        print(f"queue {q:>3}: {line}")
    print("```")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 4, int(sys.argv[3]) if len(sys.argv) > 3 else 4)
    else:
        analyse(*sys.argv[2:4])
