import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from platipy_amd import _lib
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
F = rng.standard_normal((40, 48, 56)).astype(np.float32) * 100
M = rng.standard_normal((44, 50, 52)).astype(np.float32) * 100
dF, dM = torch.from_numpy(F).cuda(), torch.from_numpy(M).cuda()
fs, ms = (56, 48, 40), (52, 50, 44)
Af, bf = np.eye(3) * 2.0, np.array([0.5, 0.5, 0.5])
Am0 = np.array([[1.9137, 0.1071, 0.0031], [-0.0813, 2.0519, 0.0207], [0.0109, 0.0043, 2.1011]])
bm0 = np.array([0.7123, -0.4057, 0.9131])
Ams = [Am0 + 0.01 * rng.standard_normal((3, 3)) for _ in range(16)]
bms = [bm0 + 0.5 * rng.standard_normal(3) for _ in range(16)]
vsize, stride = (27, 23, 19), 2
for metric in (0, 1):
    for n in (1, 3, 4, 7, 16):
        os.environ["PP_METRIC_LANES"] = "0"; _lib.reload_switches()
        a = ctx.metric_values_affine(metric, dF, fs, dM, ms, Af.ravel(), bf, Ams[:n], bms[:n], vsize, stride)
        os.environ["PP_METRIC_LANES"] = "1"; _lib.reload_switches()
        b = ctx.metric_values_affine(metric, dF, fs, dM, ms, Af.ravel(), bf, Ams[:n], bms[:n], vsize, stride)
        b2 = ctx.metric_values_affine(metric, dF, fs, dM, ms, Af.ravel(), bf, Ams[:n], bms[:n], vsize, stride)
        err = np.abs(a - b).max() / max(np.abs(a).max(), 1e-300)
        print(metric, n, "rel err", err, "repeat equal", np.array_equal(b, b2))
        if err > 1e-10:
            print(a[:3]); print(b[:3])
