#!/usr/bin/env python
"""Stress the stream-parallel atlas path for scheduling-dependent results: 4 atlases at 256x256x128 on 4 HIP streams,
repeated, each run compared bit for bit with one sequential run; repeated under environment knobs that switch single
changes off.  Usage: stress_streams.py [runs-per-setting]"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import platipy_amd as pa  # noqa: E402
from bench import synth_pair  # noqa: E402
from platipy_amd import _lib  # noqa: E402
from platipy_amd.projects.multiatlas import MUTLIATLAS_SETTINGS_DEFAULTS, run_segmentation  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shape, spacing = (128, 256, 256), (1.0, 1.0, 1.0)
device = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
nz, ny, nx = shape
x = torch.arange(nx, device=device, dtype=torch.float32).view(1, 1, nx)
y = torch.arange(ny, device=device, dtype=torch.float32).view(1, ny, 1)
z = torch.arange(nz, device=device, dtype=torch.float32).view(nz, 1, 1)
label = (((x - 0.5 * nx) / (0.2 * nx)) ** 2 + ((y - 0.5 * ny) / (0.18 * ny)) ** 2 + ((z - 0.5 * nz) / (0.25 * nz)) ** 2 < 1).to(torch.uint8)
ids = [f"{i:03d}" for i in range(4)]
atlases, target = {}, None
for i, cid in enumerate(ids):
    target, ct, _, lab = synth_pair(ctx, shape, spacing, 1234, device, warp_seed=2000 + i, label=label)
    atlases[cid] = {"CT Image": pa.Image(ct, spacing), "HEART": pa.Image(lab, spacing)}
st = copy.deepcopy(MUTLIATLAS_SETTINGS_DEFAULTS)
st["atlas_settings"]["atlas_id_list"] = ids
st["atlas_settings"]["atlas_structure_list"] = ["HEART"]
st["label_fusion_settings"]["vote_type"] = "local"
tgt = pa.Image(target, spacing)

from platipy_amd.registration import deformable as _D, utils as _U  # noqa: E402

_execute = _D.HipDemonsFilter.Execute
_need = _U._need_masks


def eager_execute(self, f, m):
    out = _execute(self, f, m)
    self._resolve()          # reads the measurements back right away: a host-device round trip per level, as before round 3
    return out


# (name, environment, eager statistics, uncached row masks, device-wide synchronize between runs)
settings = [("default", {}, False, False, False),
            ("default, device synchronised between runs", {}, False, False, True),
            ("eager statistics", {}, True, False, False),
            ("eager statistics, device synchronised between runs", {}, True, False, True),
            ("uncached row masks", {}, False, True, False),
            ("round-2 linear stage", {"PP_NO_FIXED_SAMPLES": "1", "PP_METRIC_BLOCKS": "1024"}, False, False, False)]
for name, env, eager, uncached, sync in settings:
    for k in ("PP_NO_FIXED_SAMPLES", "PP_METRIC_BLOCKS", "PP_FIR_MARCH_SP"):
        os.environ.pop(k, None)
    os.environ.update(env)
    _lib.reload_switches()      # (the library reads its PP_* switches once)
    _D.HipDemonsFilter.Execute = eager_execute if eager else _execute
    if uncached and not hasattr(_need, "__wrapped__"):
        print(f"{name}: skipped (registration/utils.py::_need_masks is no longer a cached wrapper)", flush=True)
        continue
    _U._need_masks = _need.__wrapped__ if uncached else _need
    torch.cuda.synchronize()
    run_segmentation(tgt, st, atlases=atlases, streams_per_gpu=1)       # warm-up: caches, workspaces
    torch.cuda.synchronize()
    results = []
    for r in range(runs):
        streams = 4 if r % 3 else 1                                    # sequential runs interleaved with stream-parallel ones
        seg, prob = run_segmentation(tgt, st, atlases=atlases, streams_per_gpu=streams)
        results.append((streams, prob["HEART"].numpy().copy()))
        if sync:
            torch.cuda.synchronize()
    # the majority result is the reference
    keys = [hash(p.tobytes()) for _, p in results]
    major = max(set(keys), key=keys.count)
    ref = next(p for (_, p), k in zip(results, keys) if k == major)
    bad = [(st_, float(np.abs(p - ref).max())) for (st_, p), k in zip(results, keys) if k != major]
    print(f"{name}: {len(bad)}/{runs} runs deviate from the majority result: {[(s_, round(d, 4)) for s_, d in bad]}", flush=True)
