#!/usr/bin/env python
"""How coherent are kernel B's gathers on a real field?  A thread samples the x-neighbours (2l, y) and (2l + 1, y); a
wavefront holds two rows of 32 such pairs.  Counts, on config 2's final field: pairs whose y and z cells agree (one
16-byte load per row would serve both samples), and wavefronts in which every pair does."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import platipy_amd as pa  # noqa: E402
from bench import synth_pair  # noqa: E402
from platipy_amd import _lib  # noqa: E402

ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
fixed, moving, _ = synth_pair(ctx, (256, 512, 512), (1.0, 1.0, 1.0), 1234, torch.device("cuda", 0))
fi, mi = pa.Image(fixed, (1.0, 1.0, 1.0)), pa.Image(moving, (1.0, 1.0, 1.0))
_, _, dvf = pa.registration.fast_symmetric_forces_demons_registration(fi, mi)
d = dvf.tensor                                   # [3][Z][Y][X], mm = voxels here
print("field: max |d| %.2f, rms %.2f voxels" % (float(d.abs().max()), float((d * d).sum(0).mean().sqrt())))
fl = torch.floor(d)
a, b = fl[:, :, :, 0::2], fl[:, :, :, 1::2]      # the pair's two samples
row_ok = (a[1] == b[1]) & (a[2] == b[2])
x_ok = ((b[0] - a[0]) >= -1) & ((b[0] - a[0]) <= 1)      # cell of B = cell of A + {0, 1, 2}
ok = row_ok & x_ok
print("pairs whose y and z cells agree: %.4f; and x cells within one quad: %.4f" % (float(row_ok.float().mean()), float(ok.float().mean())))
Z, Y, XP = ok.shape
w = ok[:, : Y // 2 * 2, : XP // 32 * 32].reshape(Z, Y // 2, 2, XP // 32, 32)      # wavefront = 2 rows x 32 pairs
wave_ok = w.all(dim=4).all(dim=2)
print("wavefronts in which every pair agrees: %.4f" % float(wave_ok.float().mean()))
for k in (0, 1, 2):
    g = (d[k][:, :, 1:] - d[k][:, :, :-1]).abs()
    print("component %d: mean |d/dx| %.4f voxels per voxel" % (k, float(g.mean())))
