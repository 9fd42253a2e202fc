#!/usr/bin/env python
"""Cost of iterative atlas removal at scale: 16 atlases on one GPU, cProfile of the run_iar part."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from platipy_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
per_gpu = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pr = cProfile.Profile()
pr.enable()
dt, dice, removed = bench.multi_atlas_streams_leg(ctx, (256, 512, 512), (1.0, 1.0, 1.0), dev, 0, 1, per_gpu=per_gpu, streams=4)
pr.disable()
print(f"{per_gpu} atlases on one GPU, 4 streams, IAR on: {dt:.3f} s ({60 * per_gpu / dt:.0f} atlases/min), dice {dice:.4f}, removed {removed}")
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats("iar|fusion|projection|numpy|median|curve_fit|histogram", 30)
