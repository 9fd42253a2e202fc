"""How far does the fp64 ORACLE's own 128^3 registration move when its input is perturbed by one fp32 ulp?
(the yardstick for the fp32-vs-fp64 tolerance of config 1: differences of that size are the algorithm's
conditioning -- thresholded updates at steep edges, default-0 warps into a -1000 background -- not kernel error)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle as O
from tests.helpers import phantom, random_dvf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
shape, spacing = (n, n, n), (1.0, 1.0, 1.0)
fix = phantom(shape, seed=1234, n_blobs=12)
dv = random_dvf(shape, spacing, seed=1236, max_mm=6.0, cells=8)
mov = O.warp_image(O.Vol(phantom(shape, seed=1234, n_blobs=12, noise=0), spacing), dv.astype(np.float64), edge_value=-1000.0).arr
mov = (mov + np.random.default_rng(1237).normal(0, 5, size=shape)).astype(np.float32)
t0 = time.time()
_, d0, _ = O.fast_symmetric_forces_demons_registration(O.Vol(fix, spacing), O.Vol(mov, spacing))
print("oracle seconds", time.time() - t0)
mov2 = np.nextafter(mov, np.float32(np.inf)).astype(np.float32)     # +1 ulp everywhere
_, d1, _ = O.fast_symmetric_forces_demons_registration(O.Vol(fix, spacing), O.Vol(mov2, spacing))
err = np.abs(d1.arr - d0.arr)
print("median", np.median(err), "p99", np.quantile(err, 0.99), "rms", np.sqrt((err ** 2).mean()), "inner max", err[:, 6:-6, 6:-6, 6:-6].max(), "max", err.max(),
      "frac>0.01", (err > 0.01).mean(), "frac>0.05", (err > 0.05).mean())
