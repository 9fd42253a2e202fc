#!/usr/bin/env python
"""Per-level cost of the pipelines' two linear registrations (quick similarity at shrink 8; affine 16/8/4) at 512x512x256."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import platipy_amd as pa  # noqa: E402
from bench import synth_pair  # noqa: E402
from platipy_amd import _lib  # noqa: E402
from platipy_amd.projects.multiatlas import MUTLIATLAS_SETTINGS_DEFAULTS, QUICK_REG_SETTINGS  # noqa: E402
from platipy_amd.registration import linear  # noqa: E402

if len(sys.argv) > 1:      # another build of the library (A/B measurements)
    _lib._DLL = _lib.load(os.path.abspath(sys.argv[1]))
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
fixed, moving, _ = synth_pair(ctx, (256, 512, 512), (1.0, 1.0, 1.0), 1234, torch.device("cuda", 0), warp_seed=2000)
fi, mi = pa.Image(fixed, (1.0, 1.0, 1.0)), pa.Image(moving, (1.0, 1.0, 1.0))
orig = linear._optimise_level_native
log = []


def timed(ctx_, ms, model, params, opt, n_it, verbose, *rest):
    torch.cuda.synchronize()
    e0 = ms.evaluations
    t0 = time.perf_counter()
    out = orig(ctx_, ms, model, params, opt, n_it, verbose, *rest)
    log.append((tuple(ms.vsize), ms.stride, time.perf_counter() - t0, ms.evaluations - e0))
    return out


linear._optimise_level_native = timed
orig_grad = linear.itk_moving_gradient


def timed_grad(ctx_, moving_):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = orig_grad(ctx_, moving_)
    torch.cuda.synchronize()
    print(f"   filtered gradient image: {(time.perf_counter() - t0) * 1e3:.2f} ms")
    return out


linear.itk_moving_gradient = timed_grad
cases = [(name, dict(kw, itk_sampling=itk)) for itk in (True, False)
         for name, kw in (("quick", QUICK_REG_SETTINGS), ("affine", MUTLIATLAS_SETTINGS_DEFAULTS["linear_registration_settings"]))]
for name, kw in cases:
    pa.registration.linear_registration(fi, mi, **kw)
    log.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pa.registration.linear_registration(fi, mi, **kw)
    torch.cuda.synchronize()
    print(name, "itk_sampling", kw["itk_sampling"], "total", round((time.perf_counter() - t0) * 1e3, 2), "ms")
    for vsize, stride, dt, ev in log:
        print(f"   level vsize {vsize} stride {stride}: {dt * 1e3:7.2f} ms, {ev} evaluations, {dt * 1e6 / max(ev, 1):.1f} us/evaluation")
