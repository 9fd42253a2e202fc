#!/usr/bin/env python
"""Sizes and wall time of every demons level and linear level inside one atlas chain (bench.py's multi_atlas leg)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import multi_atlas_leg, synth_pair  # noqa: E402
from platipy_amd import _lib  # noqa: E402
from platipy_amd.registration import deformable, linear  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
fixed, moving, geom = synth_pair(ctx, (256, 512, 512), (1.0, 1.0, 1.0), 1234, dev)
log = []
orig_exec = deformable.HipDemonsFilter.Execute


def exec_timed(self, f, m):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = orig_exec(self, f, m)
    torch.cuda.synchronize()
    log.append(("demons", tuple(f.GetSize()), self.GetElapsedIterations(), time.perf_counter() - t0))
    return out


orig_lin = linear._optimise_level_native


def lin_timed(ctx_, ms, model, params, opt, n_it, verbose):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = orig_lin(ctx_, ms, model, params, opt, n_it, verbose)
    log.append(("linear", tuple(ms.vsize), ms.stride, time.perf_counter() - t0))
    return out


multi_atlas_leg(ctx, fixed, moving, (1.0, 1.0, 1.0), 0, 1, dev)
deformable.HipDemonsFilter.Execute = exec_timed
linear._optimise_level_native = lin_timed
torch.cuda.synchronize()
t0 = time.perf_counter()
dt, n, dice = multi_atlas_leg(ctx, fixed, moving, (1.0, 1.0, 1.0), 0, 1, dev)
print("leg (second of two inside)", dt, "dice", dice)
half = len(log) // 2
for row in log[half:]:
    print(row[0], row[1], row[2], f"{row[3] * 1e3:.2f} ms")
