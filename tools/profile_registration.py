#!/usr/bin/env python
"""Kernel-level breakdown of one whole config-2 registration (run under rocprofv3 --kernel-trace)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import platipy_amd as pa  # noqa: E402
from bench import synth_pair  # noqa: E402
from platipy_amd import _lib  # noqa: E402

ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
fixed, moving, _ = synth_pair(ctx, (256, 512, 512), (1.0, 1.0, 1.0), 1234, torch.device("cuda", 0))
fi, mi = pa.Image(fixed, (1.0, 1.0, 1.0)), pa.Image(moving, (1.0, 1.0, 1.0))
pa.registration.fast_symmetric_forces_demons_registration(fi, mi)
torch.cuda.synchronize()
print("MARK_START", time.perf_counter_ns())
t0 = time.perf_counter()
pa.registration.fast_symmetric_forces_demons_registration(fi, mi)
torch.cuda.synchronize()
print("registration_s", time.perf_counter() - t0)
