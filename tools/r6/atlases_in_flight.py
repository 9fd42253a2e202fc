#!/usr/bin/env python3
"""Atlases/min of ONE GPU against the number of atlases it is given ("run it with 8 and 16 atlases on the
one GPU and report atlases/min vs atlases in flight"): bench.multi_atlas_streams_leg at 512x512x256, pipeline defaults, 4 HIP
streams, iterative atlas selection as the leg has it (on from 8 atlases) and, for those, with it forced off -- the difference is
what the selection costs.   python tools/r6/atlases_in_flight.py [counts ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from platipy_amd import _lib  # noqa: E402
from platipy_amd.projects import multiatlas  # noqa: E402

counts = [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8, 16]
dev = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
real = multiatlas.atlas_pipeline


def without_selection(img, settings, *a, **k):
    settings = dict(settings)
    settings["iar_settings"] = dict(settings["iar_settings"], reference_structure=False)
    return real(img, settings, *a, **k)


for n in counts:
    streams = min(4, n)
    dt, dice, removed = bench.multi_atlas_streams_leg(ctx, (256, 512, 512), (1.0, 1.0, 1.0), dev, 0, 1, per_gpu=n, streams=streams)
    line = f"{n:2d} atlases, {streams} streams: {dt:.4f} s = {60 * n / dt:6.0f} atlases/min, dice {dice:.4f}"
    if n >= 8:
        multiatlas.atlas_pipeline = without_selection
        try:
            dt0, dice0, _ = bench.multi_atlas_streams_leg(ctx, (256, 512, 512), (1.0, 1.0, 1.0), dev, 0, 1, per_gpu=n, streams=streams)
        finally:
            multiatlas.atlas_pipeline = real
        line += f"; removed {removed}; without atlas selection {dt0:.4f} s = {60 * n / dt0:6.0f} atlases/min -> selection costs {1e3 * (dt - dt0):.0f} ms"
    print(line, flush=True)
