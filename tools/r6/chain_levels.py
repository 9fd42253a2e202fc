#!/usr/bin/env python
"""The demons levels of one atlas chain (bench.py's multi_atlas leg): grid, iterations run, wall time of each Execute (device
synchronised) and which fused-kernel generation served it (the context's per-kernel profiler)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import multi_atlas_leg, synth_pair  # noqa: E402
from platipy_amd import _lib, runtime  # noqa: E402
from platipy_amd.registration import deformable  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
fixed, moving, geom = synth_pair(ctx, (256, 512, 512), (1.0, 1.0, 1.0), 1234, dev)
multi_atlas_leg(ctx, fixed, moving, (1.0, 1.0, 1.0), 0, 1, dev)          # warm
orig = deformable.HipDemonsFilter.Execute
rows = []


def timed(self, f, m):
    pctx = runtime.context(f.tensor.device)
    pctx.profile_enable(True, every=10)
    pctx.profile_read()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = orig(self, f, m)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    names = {k: v[0] for k, v in pctx.profile_read().items() if v[0] > 0}
    pctx.profile_enable(False)
    rows.append((tuple(f.tensor.shape)[::-1], self.GetElapsedIterations(), dt * 1e3, sorted(names)))
    return out


deformable.HipDemonsFilter.Execute = timed
for cube in (None, "0"):
    if cube is None:
        os.environ.pop("PP_FUSED_CUBE", None)
    else:
        os.environ["PP_FUSED_CUBE"] = cube
    _lib.reload_switches()
    rows.clear()
    dt, n, dice = multi_atlas_leg(ctx, fixed, moving, (1.0, 1.0, 1.0), 0, 1, dev)
    print(f"PP_FUSED_CUBE={cube}: chain {dt * 1e3:.2f} ms (with the per-level synchronisation), dice {dice:.4f}")
    for shape, it, ms, names in rows[len(rows) // 2:]:
        print(f"   level {shape}: {it} iterations, {ms:.2f} ms = {1e3 * ms / max(it, 1):.1f} us per iteration; kernels {names}")
