#!/usr/bin/env python
"""Latency of the linear-registration metric entry points at the three level sizes of the pipelines' affine stage
(512x512x256 images, shrink 16/8/4, sampling 0.75 -> stride 2)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platipy_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
# optional argument: another build of the library (A/B measurements)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream, _lib.load(sys.argv[1]) if len(sys.argv) > 1 else None)
F = torch.randn((256, 512, 512), device=dev)
M = torch.randn((256, 512, 512), device=dev)
fs = (512, 512, 256)
for shrink in (16, 8, 4):
    vs = (512 // shrink, 512 // shrink, 256 // shrink)
    A = np.eye(3) * shrink
    b = np.full(3, (shrink - 1) / 2.0)
    Am = A + 0.01
    cands = [Am + 0.0015 * k for k in range(16)]      # a line search's candidates: distinct, a fraction of a voxel apart
    for name, fn, reps in (
        ("single value+grad", lambda: ctx.meansq_affine(F, fs, M, fs, A.ravel(), b, Am.ravel(), b, vs, 2), 200),
        ("batch K=1", lambda: ctx.metric_values_affine(0, F, fs, M, fs, A.ravel(), b, [Am], [b], vs, 2), 200),
        ("batch K=4", lambda: ctx.metric_values_affine(0, F, fs, M, fs, A.ravel(), b, cands[:4], [b] * 4, vs, 2), 200),
        ("batch K=8", lambda: ctx.metric_values_affine(0, F, fs, M, fs, A.ravel(), b, cands[:8], [b] * 8, vs, 2), 200),
        ("batch K=16", lambda: ctx.metric_values_affine(0, F, fs, M, fs, A.ravel(), b, cands[:16], [b] * 16, vs, 2), 200),
    ):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        dt = (time.perf_counter() - t0) / reps
        print(f"shrink {shrink:2d} vsize {vs}: {name:18s} {dt * 1e6:8.1f} us")
