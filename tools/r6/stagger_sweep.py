#!/usr/bin/env python3
"""Atlases/min of one GPU against the schedule of its atlas chains.

    python tools/r6/stagger_sweep.py [reps]

bench.multi_atlas_streams_leg (512x512x256, pipeline defaults) with 4 / 8 / 16 atlases on 4 streams (and 8 on 8), for
  lockstep     : projects.multiatlas.STAGGER = False (round 5's schedule)
  turnstile    : STAGGER, ENTRY_SLOTS = 0 (only the throughput-bound phase is serialised)
  slots=1 / 2  : STAGGER, ENTRY_SLOTS = 1 / 2 (chains admitted to their linear stage one / two at a time)
`reps` alternating rounds; prints seconds, atlases/min and the Dice of the fused label per cell."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from platipy_amd import _lib  # noqa: E402
from platipy_amd.projects import multiatlas  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
shapes = [(4, 4), (8, 4), (16, 4), (8, 8)]
if len(sys.argv) > 2:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]]
dev = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
schedules = [("lockstep", False, 0), ("turnstile", True, 0), ("slots=1", True, 1), ("slots=2", True, 2)]
for atlases, streams in shapes:
    for rep in range(reps):
        for name, stagger, slots in schedules:
            multiatlas.STAGGER, multiatlas.ENTRY_SLOTS = stagger, slots
            dt, dice, removed = bench.multi_atlas_streams_leg(ctx, (256, 512, 512), (1.0, 1.0, 1.0), dev, 0, 1, per_gpu=atlases, streams=streams)
            print(f"{atlases:2d} atlases on {streams} streams, {name:9s}: {dt:.4f} s = {60 * atlases / dt:6.0f} atlases/min, dice {dice:.4f}"
                  f"{' removed ' + str(removed) if removed else ''}", flush=True)
