"""Where does a config-2 registration's wall time go beyond its kernels?  (round 4: 16.1 ms of kernels in a 16.1 ms device span,
17.3-17.7 ms wall.)  Prints the host's enqueue time, the stream's own elapsed time and the wall time, then the same with every
C-ABI call stamped (first / last launch relative to entry / return)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import synth_pair  # noqa: E402
from platipy_amd import _lib  # noqa: E402
from platipy_amd.image import Image  # noqa: E402
from platipy_amd.registration.deformable import fast_symmetric_forces_demons_registration as reg  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
shape, spacing = (256, 512, 512), (1.0, 1.0, 1.0)
fixed, moving, _ = synth_pair(ctx, shape, spacing, 1234, dev)
fi, mi = Image(fixed, spacing), Image(moving, spacing)
for _ in range(2):
    reg(fi, mi)
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    out = reg(fi, mi)
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"run {rep}: host enqueue {1e3 * (t1 - t0):.3f} ms | stream elapsed {e0.elapsed_time(e1):.3f} ms | wall {1e3 * (t2 - t0):.3f} ms")

# stamp every ABI call
stamps = []
lib = _lib.Context
orig = {}
for name in dir(lib):
    fn = getattr(lib, name)
    if callable(fn) and not name.startswith("_") and name not in ("close",):
        def wrap(fn=fn, name=name):
            def w(self, *a, **k):
                stamps.append((time.perf_counter(), name))
                return fn(self, *a, **k)
            return w
        orig[name] = fn
        setattr(lib, name, wrap())
torch.cuda.synchronize()
t0 = time.perf_counter()
reg(fi, mi)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"stamped run: {len(stamps)} ABI calls; first at +{1e3 * (stamps[0][0] - t0):.3f} ms ({stamps[0][1]}), last at +{1e3 * (stamps[-1][0] - t0):.3f} ms ({stamps[-1][1]}), "
      f"return at +{1e3 * (t1 - t0):.3f} ms, synchronised at +{1e3 * (t2 - t0):.3f} ms")
prev = t0
gaps = sorted(((s[0] - p, s[1], p_name) for (s, p, p_name) in zip(stamps, [t0] + [s[0] for s in stamps[:-1]], ["entry"] + [s[1] for s in stamps[:-1]])), reverse=True)[:8]
for g, name, before in gaps:
    print(f"  host gap {1e3 * g:.3f} ms before {name} (after {before})")
