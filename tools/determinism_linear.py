#!/usr/bin/env python
"""Where does a non-reproducible affine registration first leave the majority trajectory?  Full-precision per-iteration
metric values of every optimiser level (pp_linear_optimize_f32's history), N repeats on the same pair."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import platipy_amd as pa  # noqa: E402
from bench import synth_pair  # noqa: E402
from platipy_amd import _lib  # noqa: E402
from platipy_amd.projects.multiatlas import MUTLIATLAS_SETTINGS_DEFAULTS  # noqa: E402
from platipy_amd.registration import linear  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
shape, spacing = (128, 256, 256), (1.0, 1.0, 1.0)
device = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
fixed, moving, _ = synth_pair(ctx, shape, spacing, 1234, device, warp_seed=2001)
fi, mi = pa.Image(fixed, spacing), pa.Image(moving, spacing)
kw = MUTLIATLAS_SETTINGS_DEFAULTS["linear_registration_settings"]

trace = []
orig = _lib.Context.linear_optimize


def spy(self, *a, **k):
    k["history"] = 200
    out, stats, history = orig(self, *a, **k)
    trace.append((np.asarray(history, dtype=np.float64).copy(), np.asarray(out, dtype=np.float64).copy(), stats.evaluations))
    return out, stats, history


_lib.Context.linear_optimize = spy
runs = []
for _ in range(N):
    trace.clear()
    pa.registration.linear_registration(fi, mi, **kw)
    torch.cuda.synchronize()
    runs.append([(h.copy(), p.copy(), e) for h, p, e in trace])
keys = [hash(b"".join(h.tobytes() + p.tobytes() for h, p, _ in r)) for r in runs]
major = max(set(keys), key=keys.count)
ref = runs[keys.index(major)]
print(f"{sum(k != major for k in keys)}/{N} runs leave the majority trajectory; levels per run {len(ref)}, iterations {[len(h) for h, _, _ in ref]}")
for i, (r, k) in enumerate(zip(runs, keys)):
    if k == major:
        continue
    for lv, ((h, p, e), (h0, p0, e0)) in enumerate(zip(r, ref)):
        n = min(len(h), len(h0))
        d = np.nonzero(h[:n] != h0[:n])[0]
        if len(d) or len(h) != len(h0) or not np.array_equal(p, p0):
            j = int(d[0]) if len(d) else n
            print(f"run {i}: level {lv}: first differing iteration {j} of {len(h0)} (this run {len(h)}), "
                  f"value {h[j] if j < len(h) else None!r} vs {h0[j] if j < len(h0) else None!r}, "
                  f"rel diff {abs(h[j] - h0[j]) / abs(h0[j]) if j < n else float('nan'):.3g}; evaluations {e} vs {e0}; "
                  f"max param diff {float(np.abs(p - p0).max()):.3g}")
            break
