#!/usr/bin/env python
"""Where does one atlas chain spend its wall time?  (cProfile of run_segmentation at the bench size.)"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import multi_atlas_leg, synth_pair  # noqa: E402
from platipy_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
fixed, moving, geom = synth_pair(ctx, (256, 512, 512), (1.0, 1.0, 1.0), 1234, dev)
pr = cProfile.Profile()
pr.enable()
dt, n, dice = multi_atlas_leg(ctx, fixed, moving, (1.0, 1.0, 1.0), 0, 1, dev)
pr.disable()
print("timed run", dt, "dice", dice)
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
