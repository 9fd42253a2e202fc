#!/usr/bin/env python3
"""Line-search speculation depth (candidates per probe launch = 2^depth - 1) against the chains' wall time, with ITK's sampling.

    python tools/r6/speculation_sweep.py [reps]

Round 3 chose depth 4 on a solo affine stage with lattice samples (32.4 / 34.0 / 38.0 / 65.3 ms at depth 4 / 3 / 2 / 1).  With
jittered samples a candidate costs more, and four chains side by side are bound by the metric kernels' total work, of which a
depth-4 launch wastes 11 of 15 candidates: bench.multi_atlas_leg (one chain) and bench.multi_atlas_streams_leg (4 on 4 streams)
per depth, `reps` alternating rounds."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from platipy_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
shape, spacing = (256, 512, 512), (1.0, 1.0, 1.0)
fixed = bench.synth_pair(ctx, shape, spacing, 1234, dev)[0]
for rep in range(reps):
    for depth in (4, 3, 2):
        os.environ["PP_LINE_SEARCH_SPECULATION"] = str(depth)
        one = bench.multi_atlas_leg(ctx, fixed, None, spacing, 0, 1, dev)
        four = bench.multi_atlas_streams_leg(ctx, shape, spacing, dev, 0, 1, per_gpu=4, streams=4)
        print(f"depth {depth}: one chain {one[0]:.4f} s (dice {one[2]:.4f}); 4 chains on 4 streams {four[0]:.4f} s = "
              f"{240 / four[0]:.0f} atlases/min, dice {four[1]:.4f}", flush=True)
