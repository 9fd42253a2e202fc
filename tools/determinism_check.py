#!/usr/bin/env python
"""Which stage of an atlas chain is not reproducible run to run?  Each stage repeated N times on the same inputs (one
stream, device synchronised between repeats), outputs compared bit for bit with the majority result."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import platipy_amd as pa  # noqa: E402
from bench import synth_pair  # noqa: E402
from platipy_amd import _lib  # noqa: E402
from platipy_amd.projects.multiatlas import MUTLIATLAS_SETTINGS_DEFAULTS, QUICK_REG_SETTINGS  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
NA = int(sys.argv[2]) if len(sys.argv) > 2 else 3 * N   # the affine stage (line search: many mailbox round trips) gets more repeats
shape, spacing = (128, 256, 256), (1.0, 1.0, 1.0)
device = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
fixed, moving, _ = synth_pair(ctx, shape, spacing, 1234, device, warp_seed=2001)
fi, mi = pa.Image(fixed, spacing), pa.Image(moving, spacing)
st = MUTLIATLAS_SETTINGS_DEFAULTS


def report(name, outs):
    keys = [hash(o.tobytes()) for o in outs]
    major = max(set(keys), key=keys.count)
    ref = outs[keys.index(major)]
    bad = [float(np.abs(o.astype(np.float64) - ref).max()) for o, k in zip(outs, keys) if k != major]
    print(f"{name}: {len(bad)}/{len(outs)} repeats deviate {['%.3g' % b for b in bad]}", flush=True)


def params_of(tfm):
    t = tfm.transforms[-1] if hasattr(tfm, "transforms") else tfm
    return np.asarray(t.GetParameters(), dtype=np.float64)


outs = []
for _ in range(N):
    _, tfm = pa.registration.linear_registration(fi, mi, **QUICK_REG_SETTINGS)
    torch.cuda.synchronize()
    outs.append(params_of(tfm))
report("quick linear registration (parameters)", outs)
outs = []
for _ in range(NA):
    img, tfm = pa.registration.linear_registration(fi, mi, **st["linear_registration_settings"])
    torch.cuda.synchronize()
    outs.append(params_of(tfm))
report("affine linear registration (parameters)", outs)
lin_img = img
outs = []
for _ in range(N):
    _, _, dvf = pa.registration.fast_symmetric_forces_demons_registration(fi, lin_img, **st["deformable_registration_settings"])
    torch.cuda.synchronize()
    outs.append(dvf.tensor.cpu().numpy())
report("demons registration, pipeline settings (field)", outs)
outs = []
for _ in range(N):
    _, _, dvf = pa.registration.fast_symmetric_forces_demons_registration(fi, lin_img, resolution_staging=[4, 2, 1], iteration_staging=[10, 10, 5])
    torch.cuda.synchronize()
    outs.append(dvf.tensor.cpu().numpy())
report("demons registration, shrink 4/2/1 (field)", outs)
