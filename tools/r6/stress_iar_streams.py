#!/usr/bin/env python
"""Scheduling-dependent results in config 5's shape WITH atlas selection: tests/test_fullsize_oracle.py's 8-atlas job at
256x256x128 (two displaced labels, iterative atlas removal), run once sequentially and then `runs` times on 4 HIP streams;
every stream run is compared with the sequential one -- atlases removed, fused mask and probability, every atlas's
propagated image, label and weight map -- and the first deviating stage is named.

    python tools/r6/stress_iar_streams.py [runs]        # PP_FUSED_CUBE=0 in the environment: the marching kernels only"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from platipy_amd import _lib  # noqa: E402
from platipy_amd.projects.multiatlas import run_segmentation  # noqa: E402
from tests.test_fullsize_oracle import _atlas_job  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
ids, atlases, target, label, st = _atlas_job(ctx, (128, 256, 256), 8, wrong=("002", "005"))
st["iar_settings"].update({"reference_structure": "HEART", "min_best_atlases": 4})


def snapshot(streams):
    res, prob, aset = run_segmentation(target, st, atlases=atlases, streams_per_gpu=streams, return_atlas_set=True)
    out = {"removed": sorted(run_segmentation.last_iar_removed), "mask": res["HEART"].tensor.clone(), "prob": prob["HEART"].tensor.clone()}
    for cid in ids:
        for stage in ("RIR", "DIR"):
            for key, img in aset.get(cid, {}).get(stage, {}).items():
                if hasattr(img, "tensor"):
                    out[f"{cid}/{stage}/{key}"] = img.tensor.clone()
    return out


ref = snapshot(1)
print("sequential: removed", ref["removed"], "keys", len(ref), flush=True)
bad = 0
for r in range(runs):
    got = snapshot(4)
    diffs = []
    for k, v in ref.items():
        if k not in got:
            diffs.append(f"{k}: missing")
        elif k == "removed":
            if v != got[k]:
                diffs.append(f"removed {got[k]}")
        elif v.shape != got[k].shape or not torch.equal(v, got[k]):
            n = int((v != got[k]).sum()) if v.shape == got[k].shape else -1
            diffs.append(f"{k}: {n} elements differ")
    bad += 1 if diffs else 0
    print(f"run {r:2d}: {'equal' if not diffs else '; '.join(diffs[:8])}", flush=True)
print(f"deviating runs: {bad} of {runs}")
