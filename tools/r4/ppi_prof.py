#!/usr/bin/env python
"""process_probability_image on a chain-sized crop (300 x 300 x 250 probability volume), to be run under rocprofv3."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import platipy_amd as pa  # noqa: E402
from platipy_amd.label.fusion import process_probability_image  # noqa: E402

nz, ny, nx = 250, 300, 300
dev = torch.device("cuda", 0)
x = torch.arange(nx, device=dev, dtype=torch.float32).view(1, 1, nx)
y = torch.arange(ny, device=dev, dtype=torch.float32).view(1, ny, 1)
z = torch.arange(nz, device=dev, dtype=torch.float32).view(nz, 1, 1)
r = ((x - 150) / 60) ** 2 + ((y - 150) / 55) ** 2 + ((z - 125) / 70) ** 2
prob = torch.clamp(1.2 - r, 0, 1) + 0.02 * torch.rand((nz, ny, nx), device=dev) * (r < 1.3)
img = pa.Image(prob.contiguous(), (1.0, 1.0, 1.0))
for _ in range(3):
    out = process_probability_image(img, 0.5)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    out = process_probability_image(img, 0.5)
torch.cuda.synchronize()
print("process_probability_image", (time.perf_counter() - t0) / 10 * 1e3, "ms; mask voxels", int(out.tensor.sum()))
