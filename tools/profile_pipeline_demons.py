#!/usr/bin/env python
"""Where one pipeline-default demons registration (isotropic 6/3/1.5 mm x 150/125/100, multiatlas/run.py:76-84)
spends its time at 512x512x256: stage-level wall clock with a device sync after each stage."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import platipy_amd as pa  # noqa: E402
from bench import synth_pair  # noqa: E402
from platipy_amd import _lib  # noqa: E402
from platipy_amd.registration import deformable, utils  # noqa: E402

ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
fixed, moving, _ = synth_pair(ctx, (256, 512, 512), (1.0, 1.0, 1.0), 1234, torch.device("cuda", 0))
fi, mi = pa.Image(fixed, (1.0, 1.0, 1.0)), pa.Image(moving, (1.0, 1.0, 1.0))
kw = dict(isotropic_resample=True, resolution_staging=[6, 3, 1.5], iteration_staging=[150, 125, 100], smoothing_sigmas=[0, 0, 0])
pa.registration.fast_symmetric_forces_demons_registration(fi, mi, **kw)
torch.cuda.synchronize()

acc = {}


def wrap(mod, name):
    fn = getattr(mod, name)

    def timed(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize()
        acc.setdefault(name, []).append(time.perf_counter() - t0)
        return out

    setattr(mod, name, timed)


for mod, name in ((deformable, "smooth_and_resample"), (deformable, "resample_field"), (deformable, "resample_image"),
                  (deformable, "apply_transform")):
    if hasattr(mod, name):
        wrap(mod, name)
orig_exec = deformable.HipDemonsFilter.Execute


def exec_timed(self, f, m):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = orig_exec(self, f, m)
    torch.cuda.synchronize()
    acc.setdefault("Execute", []).append((time.perf_counter() - t0, f.GetSize(), self.GetElapsedIterations()))
    return out


deformable.HipDemonsFilter.Execute = exec_timed
t0 = time.perf_counter()
pa.registration.fast_symmetric_forces_demons_registration(fi, mi, **kw)
torch.cuda.synchronize()
print("total (with per-stage syncs)", time.perf_counter() - t0)
for k, v in acc.items():
    if k == "Execute":
        for dt, size, its in v:
            print(f"  Execute {size} iterations {its}: {dt * 1e3:.2f} ms  ({dt * 1e3 / max(its, 1):.3f} ms/iter)")
    else:
        print(f"  {k}: n={len(v)} total {sum(v) * 1e3:.2f} ms  each {[round(x * 1e3, 2) for x in v]}")
