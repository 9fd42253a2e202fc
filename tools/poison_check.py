#!/usr/bin/env python
"""Uninitialised-read hunt: one sequential atlas chain with every torch.empty / empty_like buffer pre-filled with NaN
(and, under PP_POISON_WS=1, the library's scratch too) against the same chain run normally.  Any difference, or any NaN
in the outputs, is a read of memory nobody wrote."""
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

shape, spacing = (128, 256, 256), (1.0, 1.0, 1.0)
device = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
nz, ny, nx = shape
x = torch.arange(nx, device=device, dtype=torch.float32).view(1, 1, nx)
y = torch.arange(ny, device=device, dtype=torch.float32).view(1, ny, 1)
z = torch.arange(nz, device=device, dtype=torch.float32).view(nz, 1, 1)
label = (((x - 0.5 * nx) / (0.2 * nx)) ** 2 + ((y - 0.5 * ny) / (0.18 * ny)) ** 2 + ((z - 0.5 * nz) / (0.25 * nz)) ** 2 < 1).to(torch.uint8)
ids = ["000", "001"]
atlases, target = {}, None
for i, cid in enumerate(ids):
    target, ct, _, lab = synth_pair(ctx, shape, spacing, 1234, device, warp_seed=2000 + i, label=label)
    atlases[cid] = {"CT Image": pa.Image(ct, spacing), "HEART": pa.Image(lab, spacing)}
st = copy.deepcopy(MUTLIATLAS_SETTINGS_DEFAULTS)
st["atlas_settings"]["atlas_id_list"] = ids
st["atlas_settings"]["atlas_structure_list"] = ["HEART"]
st["label_fusion_settings"]["vote_type"] = "local"
tgt = pa.Image(target, spacing)

# also the registration alone (config 2's shape of call), which is where most torch.empty buffers live
fi, mi = tgt, atlases["001"]["CT Image"]


def everything():
    seg, prob = run_segmentation(tgt, st, atlases=atlases, streams_per_gpu=1)
    _, _, dvf = pa.registration.fast_symmetric_forces_demons_registration(fi, mi, resolution_staging=[8, 4, 1], iteration_staging=[5, 5, 3])
    _, tfm = pa.registration.linear_registration(fi, mi, **MUTLIATLAS_SETTINGS_DEFAULTS["linear_registration_settings"])
    torch.cuda.synchronize()
    return prob["HEART"].numpy().copy(), dvf.tensor.cpu().numpy().copy(), np.asarray(tfm.GetParameters() if hasattr(tfm, "GetParameters") else tfm.transforms[-1].GetParameters())


ref = everything()
ref2 = everything()
print("plain repeat: prob", float(np.abs(ref[0] - ref2[0]).max()), "dvf", float(np.abs(ref[1] - ref2[1]).max()), "affine", float(np.abs(ref[2] - ref2[2]).max()))

_empty, _empty_like = torch.empty, torch.empty_like


def _poisoned(t):
    if t.is_floating_point() and t.device.type == "cuda":
        t.fill_(float("nan"))
    elif t.device.type == "cuda" and t.dtype in (torch.uint8, torch.int32, torch.int64):
        t.fill_(113)
    return t


torch.empty = lambda *a, **k: _poisoned(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poisoned(_empty_like(*a, **k))
got = everything()
for name, a, b in zip(("prob", "dvf", "affine"), ref, got):
    d = np.abs(a - b)
    print(f"poisoned torch.empty: {name}: NaNs {int(np.isnan(b).sum())}, max diff {float(np.nanmax(d)) if d.size else 0.0:.3g}, differing {int((d > 0).sum())}")
