#!/usr/bin/env python
"""Native optimiser against the Python loop on the pair of tests/test_linear.py::test_native_optimiser_follows_the_python_loop,
for both forms of the gradient kernel: final parameters, their difference, iteration histories."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import platipy_amd as pa  # noqa: E402
from platipy_amd.registration import linear  # noqa: E402
from tests.test_linear import _rigid_pair  # noqa: E402

shape, spacing, origin = (16, 20, 24), (1.5, 1.5, 2.5), (-30.0, -20.0, 10.0)
fix, mov, _ = _rigid_pair(pa, shape, spacing, origin, angle=0.06, shift=(2.0, -1.5, 1.0))
for method in ("scaleversor", "rigid", "affine"):
    for one in ("0", "1"):
        os.environ["PP_METRIC_GRAD_ONE_LAUNCH"] = one; _lib.reload_switches()
        out = {}
        for native in (True, False):
            linear.NATIVE_OPTIMISER = native
            _, tfm = pa.registration.linear_registration(
                pa.image_from_array(fix, spacing, origin), pa.image_from_array(mov, spacing, origin), reg_method=method,
                optimiser="gradient_descent_line_search", metric="mean_squares", shrink_factors=[2, 1], smooth_sigmas=[1, 0],
                sampling_rate=0.5, number_of_iterations=6)
            out[native] = np.asarray(tfm.transforms[1].GetParameters())
        d = np.abs(out[True] - out[False])
        print(method, "one-launch" if one == "1" else "two-launch", "max |native - python| =", d.max(), "rel", (d / np.maximum(np.abs(out[False]), 1e-8)).max())
