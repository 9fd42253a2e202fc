#!/usr/bin/env python
"""Wall time (device-synchronised) of every stage function run_segmentation calls, for one atlas chain (bench.py's
multi_atlas leg).  Inclusive times; linear_registration / demons contain their own apply_transform calls."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import multi_atlas_leg, synth_pair  # noqa: E402
from platipy_amd import _lib  # noqa: E402
from platipy_amd.projects import multiatlas as ma  # noqa: E402
from platipy_amd.registration import deformable, linear, utils as rutils  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
fixed, moving, geom = synth_pair(ctx, (256, 512, 512), (1.0, 1.0, 1.0), 1234, dev)
log, depth = [], [0]


def wrap(mod, name, label=None):
    fn = getattr(mod, name)

    def timed(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        depth[0] += 1
        try:
            return fn(*a, **k)
        finally:
            torch.cuda.synchronize()
            depth[0] -= 1
            log.append((depth[0], label or name, time.perf_counter() - t0))

    setattr(mod, name, timed)


multi_atlas_leg(ctx, fixed, moving, (1.0, 1.0, 1.0), 0, 1, dev)
for name in ("linear_registration", "apply_transform", "fast_symmetric_forces_demons_registration", "compute_weight_map", "finalize_probability",
             "process_probability_image", "label_to_roi", "crop_to_roi", "paste"):
    wrap(ma, name)
for mod, name in ((deformable, "smooth_and_resample"), (deformable, "apply_transform"), (linear, "apply_transform"),
                  (deformable, "resample_field"), (deformable, "_total_field"), (linear, "smooth_and_resample")):
    if hasattr(mod, name):
        wrap(mod, name, f"  {mod.__name__.rsplit('.', 1)[-1]}.{name}")
orig_exec = deformable.HipDemonsFilter.Execute
wrap(deformable.HipDemonsFilter, "Execute", "  demons level")
wrap(linear, "_optimise_level_native", "  linear level")
wrap(linear, "itk_moving_gradient", "  linear.itk_moving_gradient")
torch.cuda.synchronize()
log.clear()
t0 = time.perf_counter()
from platipy_amd.projects.multiatlas import run_segmentation  # noqa: E402

dt, n, dice = multi_atlas_leg(ctx, fixed, moving, (1.0, 1.0, 1.0), 0, 1, dev)
# multi_atlas_leg runs a warm-up and a timed run: keep the second half
half = len(log) // 2
tot = {}
print("timed run (with syncs)", round(dt * 1e3, 2), "ms")
for d, name, t in log[half:]:
    print(f"{'  ' * d}{name:50s} {t * 1e3:8.2f} ms")
    if d == 0:
        tot[name] = tot.get(name, 0.0) + t
print("top-level sums:", {k: round(v * 1e3, 2) for k, v in tot.items()}, "sum", round(sum(tot.values()) * 1e3, 2))
