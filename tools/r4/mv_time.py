#!/usr/bin/env python
"""HIP-event time of pp_metric_values_affine_f32 on the three lattices of the pipelines' affine stage, K = 1 and 16, inside a
fixed-sample scope as the optimiser runs it.  Usage: mv_time.py [lib.so]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from platipy_amd import _lib  # noqa: E402

if len(sys.argv) > 1:
    _lib._DLL = _lib.load(os.path.abspath(sys.argv[1]))
dev = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
F = torch.randn((256, 512, 512), device=dev)
M = torch.randn((256, 512, 512), device=dev)
fs = (512, 512, 256)
for shrink in (16, 8, 4):
    vs = (512 // shrink, 512 // shrink, 256 // shrink)
    A = np.eye(3) * shrink
    b = np.full(3, (shrink - 1) / 2.0)
    Am = A + 0.01
    cands = [Am + 0.0015 * k for k in range(16)]
    for K in (1, 4, 16):
        fn = lambda: ctx.metric_values_affine(0, F, fs, M, fs, A.ravel(), b, cands[:K], [b] * K, vs, 2)  # noqa: E731
        for _ in range(3):
            fn()
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        dt = (time.perf_counter() - t0) / 200
        print(f"shrink {shrink:2d} vsize {vs}: K={K:2d} {dt * 1e6:8.1f} us per call (wall, one call in flight)")
