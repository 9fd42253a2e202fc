#!/bin/bash
set +e
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
python -m pytest tests/test_kernels.py -m gpu -x -q -k demons 2>&1 | tail -2
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from bench import synth_pair
from platipy_amd import _lib
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
fixed, moving, _ = synth_pair(ctx, (256, 512, 512), (1.0, 1.0, 1.0), 1234, torch.device("cuda", 0))
for sp in (0.9, 0.6, 0.5):
    g = _lib.make_geom((512, 512, 256), (sp, sp, 1.0))
    for name, var in (("fused", _lib.DEMONS_FUSED), ("staged", _lib.DEMONS_STAGED)):
        p = ctx.default_demons_params(); p.smooth_update = 1; p.max_rms_error = 0.0; p.variant = var
        p.sigma_d_vox[:] = [1.5 / sp, 1.5 / sp, 1.5]
        field = torch.empty((3, 256, 512, 512), device="cuda")
        p.iterations = 3; ctx.demons_execute(fixed, moving, g, p, field, want_stats=False); torch.cuda.synchronize()
        p.iterations = 20; t0 = time.perf_counter(); ctx.demons_execute(fixed, moving, g, p, field, want_stats=False); torch.cuda.synchronize()
        print(f"spacing {sp}: {name:6s} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/iter")
PY
