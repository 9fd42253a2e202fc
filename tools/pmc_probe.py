#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes: two calibration kernels with known HBM byte counts (one
16 B/lane, one 4 B/lane access pattern, as MI355X_MICROARCH.md's HBM section asks before trusting
FETCH_SIZE / WRITE_SIZE) followed by a few fused demons iterations at the bench size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_pair  # noqa: E402
from platipy_amd import _lib  # noqa: E402

nx, ny, nz = (int(v) for v in os.environ.get("PP_PROBE_SIZE", "512,512,256").split(","))
shape = (nz, ny, nx)
n = nx * ny * nz
dev = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
fixed, moving, geom = synth_pair(ctx, shape, (1.0, 1.0, 1.0), 1234, dev)
a = torch.rand(3 * n, device=dev)
b = torch.rand(3 * n, device=dev) + 1.0
c = torch.empty_like(a)
torch.cuda.synchronize()
# calibration 1: torch elementwise add, 16 B/lane loads and stores: reads 12 B/voxel... 3n*4 B, writes 3n*4 B
torch.add(a, 1.0, out=c)
# calibration 2: k_fuse_divide, 4 B/lane: reads 2 * 3n*4 B, writes 3n*4 B
ctx.fuse_divide(a, b, c, 3 * n)
torch.cuda.synchronize()
field = torch.zeros((3,) + shape, device=dev)
p = ctx.default_demons_params()
p.smooth_update = 1
p.smooth_displacement = 1
p.sigma_d_vox[:] = [1.5, 1.5, 1.5]
p.max_rms_error = 0.0
p.iterations = 4
ctx.demons_execute(fixed, moving, geom, p, field, want_stats=False)
torch.cuda.synchronize()
print("probe done", n)
