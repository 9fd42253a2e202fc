#!/usr/bin/env python
"""End-to-end runs of BASELINE.json's configs that fit one GPU, with quality checks:
  config 2: 512x512x256 pair, 3-level demons (8/4/1, 10/10/10)
  config 3: linear (rigid then affine) + demons on a 512^3 pair
  config 5 shape on one GPU: 4 atlases at 512x512x256, 1 vs 4 HIP streams, iterative atlas selection on
Prints one JSON object per config."""
import copy
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import platipy_amd as pa  # noqa: E402
from bench import synth_pair  # noqa: E402
from platipy_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, out


def mse(a, b):
    return float(((a.float() - b.float()) ** 2).mean())


which = sys.argv[1:] or ["2", "3", "5"]
if "2" in which:
    fixed, moving, _ = synth_pair(ctx, (256, 512, 512), (1.0, 1.0, 1.0), 1234, dev)
    fi, mi = pa.Image(fixed, (1.0, 1.0, 1.0)), pa.Image(moving, (1.0, 1.0, 1.0))
    t, (img, tfm, dvf) = timed(lambda: pa.registration.fast_symmetric_forces_demons_registration(fi, mi))
    print(json.dumps({"config": 2, "seconds": t, "mse_before": mse(fixed, moving), "mse_after": mse(fixed, img.tensor),
                      "dvf_max_mm": float(dvf.tensor.abs().max()), "dvf_rms_mm": float((dvf.tensor ** 2).sum(0).mean().sqrt())}))
    del fixed, moving, fi, mi, img, dvf, tfm
    torch.cuda.empty_cache()
if "3" in which:
    fixed, moving0, _ = synth_pair(ctx, (512, 512, 512), (1.0, 1.0, 1.0), 4321, dev)
    # add a known rigid misalignment on top of the deformable one
    import numpy as np

    ang = 0.05
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
    mis = pa.AffineTransform(R, (6.0, -4.0, 3.0), (255.5, 255.5, 255.5))
    m0 = pa.Image(moving0, (1.0, 1.0, 1.0))
    moving = pa.registration.apply_transform(m0, m0, mis, -1000, pa.sitkLinear)
    fi = pa.Image(fixed, (1.0, 1.0, 1.0))

    def chain():
        r_img, r_tfm = pa.registration.linear_registration(fi, moving, reg_method="rigid", shrink_factors=[16, 8, 4], smooth_sigmas=[0, 0, 0],
                                                           sampling_rate=0.75, optimiser="gradient_descent_line_search")
        a_img, a_tfm = pa.registration.linear_registration(fi, r_img, reg_method="affine", shrink_factors=[16, 8, 4], smooth_sigmas=[0, 0, 0],
                                                           sampling_rate=0.75, optimiser="gradient_descent_line_search")
        d_img, d_tfm, dvf = pa.registration.fast_symmetric_forces_demons_registration(fi, a_img)
        return r_img, a_img, d_img

    t, (r_img, a_img, d_img) = timed(chain, reps=1)
    print(json.dumps({"config": 3, "seconds": t, "mse_start": mse(fixed, moving.tensor), "mse_rigid": mse(fixed, r_img.tensor),
                      "mse_affine": mse(fixed, a_img.tensor), "mse_demons": mse(fixed, d_img.tensor)}))
    del fixed, moving0, moving, fi, r_img, a_img, d_img, m0
    torch.cuda.empty_cache()
if "5" in which:
    from platipy_amd.projects.multiatlas import MUTLIATLAS_SETTINGS_DEFAULTS, run_segmentation

    shape = (256, 512, 512)
    fixed, _, _ = synth_pair(ctx, shape, (1.0, 1.0, 1.0), 1234, dev)
    nz, ny, nx = shape
    x = torch.arange(nx, device=dev, dtype=torch.float32).view(1, 1, nx)
    y = torch.arange(ny, device=dev, dtype=torch.float32).view(1, ny, 1)
    z = torch.arange(nz, device=dev, dtype=torch.float32).view(nz, 1, 1)
    label = (((x - 0.5 * nx) / (0.2 * nx)) ** 2 + ((y - 0.5 * ny) / (0.18 * ny)) ** 2 + ((z - 0.5 * nz) / (0.25 * nz)) ** 2 < 1).to(torch.uint8)
    atlases, ids = {}, []
    for i in range(4):
        # an atlas = the same anatomy seen through its own smooth field (seed 2000 + i) + the label seen through that field
        _, mov, _, lab = synth_pair(ctx, shape, (1.0, 1.0, 1.0), 1234, dev, warp_seed=2000 + i, label=label)
        cid = f"{i:03d}"
        ids.append(cid)
        atlases[cid] = {"CT Image": pa.Image(mov, (1.0, 1.0, 1.0)), "HEART": pa.Image(lab, (1.0, 1.0, 1.0))}
    st = copy.deepcopy(MUTLIATLAS_SETTINGS_DEFAULTS)
    st["atlas_settings"]["atlas_id_list"] = ids
    st["atlas_settings"]["atlas_structure_list"] = ["HEART"]
    st["label_fusion_settings"]["vote_type"] = "local"
    target = pa.Image(fixed, (1.0, 1.0, 1.0))
    out = {"config": "5 (one GPU: 4 atlases)"}
    for streams in (1, 4):
        t, (res, _) = timed(lambda: run_segmentation(target, st, atlases=atlases, streams_per_gpu=streams), reps=1)
        out[f"seconds_streams{streams}"] = t
        out[f"atlases_per_min_streams{streams}"] = 60.0 * 4 / t
        out["dice_vs_template_label"] = float(2 * ((res["HEART"].tensor > 0) & (label > 0)).sum() / ((res["HEART"].tensor > 0).sum() + (label > 0).sum()))
    print(json.dumps(out))
