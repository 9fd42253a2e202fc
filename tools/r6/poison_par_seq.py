#!/usr/bin/env python
"""Does the selection job read memory it did not write?  The 8-atlas job of tests/test_fullsize_oracle.py (256x256x128,
iterative atlas removal) on 4 streams and then sequentially, with the library's scratch poisoned on every reservation
(PP_POISON_WS=1: NaN bytes) and torch's cached blocks filled with NaN before each run -- on a box whose memory was wiped by
an earlier process an uninitialised read returns zeros in both runs and hides; here it shows as NaN or as a difference.

    PP_POISON_WS=1 python tools/r6/poison_par_seq.py [runs]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from platipy_amd import _lib  # noqa: E402
from platipy_amd.projects.multiatlas import run_segmentation  # noqa: E402
from tests.test_fullsize_oracle import _atlas_job, _first_deviation  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
ids, atlases, target, label, st = _atlas_job(ctx, (128, 256, 256), 8, wrong=("002", "005"))
st["iar_settings"].update({"reference_structure": "HEART", "min_best_atlases": 4})


def dirty_cache():
    """Leave NaN in (nearly) ALL free device memory: whatever torch or the library allocates next -- from its caches or fresh
    from the driver -- starts out as NaN, not as the zeros of a box whose memory an earlier process's exit wiped."""
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    junk, chunk = [], 1 << 30
    while free > 6 * chunk:
        try:
            junk.append(torch.full((chunk // 4,), float("nan"), device="cuda"))
        except RuntimeError:
            break
        free -= chunk
    torch.cuda.synchronize()
    n = len(junk)
    del junk
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return n


for r in range(runs):
    print("GB filled with NaN and released:", dirty_cache(), flush=True)
    par, par_p, aset = run_segmentation(target, st, atlases=atlases, streams_per_gpu=4, return_atlas_set=True)
    rem_par = sorted(run_segmentation.last_iar_removed)
    dirty_cache()
    seq, seq_p, aset_seq = run_segmentation(target, st, atlases=atlases, streams_per_gpu=1, return_atlas_set=True)
    rem_seq = sorted(run_segmentation.last_iar_removed)
    same = torch.equal(par["HEART"].tensor, seq["HEART"].tensor)
    nan = bool(torch.isnan(par_p["HEART"].tensor).any() or torch.isnan(seq_p["HEART"].tensor).any())
    print(f"run {r}: removed {rem_par} / {rem_seq}; fused masks equal: {same}; NaN in a probability: {nan}; {_first_deviation(aset, aset_seq)[:600]}",
          flush=True)
