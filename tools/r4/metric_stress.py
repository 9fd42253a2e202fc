#!/usr/bin/env python
"""Determinism of the metric entry points under back-to-back launches that reuse the same row buffer: two inputs alternate,
every result must be bit-identical to the first one of its input (a stale row of the other input's launch would show)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from platipy_amd import _lib  # noqa: E402

ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for shape, vsize in (((64, 96, 112), (48, 40, 28)), ((256, 512, 512), (128, 128, 64))):
    F = torch.randn(shape, device="cuda") * 100
    M = torch.randn(shape, device="cuda") * 100
    fs = shape[::-1]
    sh = fs[0] / vsize[0]
    Af, bf = np.eye(3) * sh, np.full(3, (sh - 1) / 2.0)
    maps = [(Af + 0.01 * (k + 1), bf + 0.3 * k) for k in range(2)]
    cands = [[(A + 0.002 * c, b) for c in range(16)] for A, b in maps]
    ref_g, ref_v, bad_g, bad_v = {}, {}, 0, 0
    for it in range(n):
        k = it & 1
        A, b = maps[k]
        g = ctx.meansq_affine(F, fs, M, fs, Af.ravel(), bf, A.ravel(), b, vsize, 2)
        v = ctx.metric_values_affine(0, F, fs, M, fs, Af.ravel(), bf, [c[0] for c in cands[k]], [c[1] for c in cands[k]], vsize, 2)
        if k not in ref_g:
            ref_g[k], ref_v[k] = np.array(g), np.array(v)
        bad_g += not np.array_equal(np.array(g), ref_g[k])
        bad_v += not np.array_equal(np.array(v), ref_v[k])
    print(f"lattice {vsize}: {n} alternating launches, gradient mismatches {bad_g}, probe mismatches {bad_v}")
