"""Multi-atlas segmentation: the harness of platipy/imaging/projects/multiatlas/run.py:106-441
(run_segmentation), re-designed for one process per MI355X.

The reference walks its atlases in serial Python loops on one CPU.  Every atlas-to-target chain
(linear -> demons -> propagate -> weight map) touches only that atlas and the target, so here
atlas i belongs to rank i % world_size (torch.distributed, backend "nccl" = RCCL over xGMI) and,
inside a rank, to one of `streams_per_gpu` worker threads each driving its own HIP stream + pp_ctx,
so one atlas's small coarse-level kernels overlap another's.  There are exactly two exchanges:
  * the auto-crop's mean of the (<= 8) quick-registered atlas images   -> one all_reduce(SUM);
  * label fusion, the single cross-atlas step (fusion.py:263,276): each rank folds its atlases into
    sum(w) and sum(w * L_s) per structure, then ONE all_reduce(SUM) of a contiguous (2 S) x N fp32
    buffer; every rank then divides, blurs, rescales and thresholds identically.
fp32 summation order differs from the reference's left fold by a few ulp; thresholded masks are
insensitive to it except at exact ties.

Atlases are passed in memory ({atlas_id: {"CT Image": Image, "<structure>": Image, ...}}) or, as in the
reference, read from NIfTI files under settings["atlas_settings"]["atlas_path"] (platipy_amd.io; each rank reads
only its own share).  Settings keep the reference's schema (MUTLIATLAS_SETTINGS_DEFAULTS, spelling included).
"""
import copy
import logging
import os
from concurrent.futures import ThreadPoolExecutor

import torch

from .. import runtime
from ..image import Image, as_image
from ..label.fusion import compute_weight_map, finalize_probability, process_probability_image
from ..registration.deformable import fast_symmetric_forces_demons_registration
from ..registration.linear import linear_registration
from ..registration.utils import apply_transform
from ..transform import sitkLinear, sitkNearestNeighbor
from ..utils.crop import crop_to_roi, label_to_roi, paste

logger = logging.getLogger(__name__)

ATLAS_PATH = os.environ.get("ATLAS_PATH", "/atlas")

MUTLIATLAS_SETTINGS_DEFAULTS = {
    "atlas_settings": {
        "atlas_id_list": ["03"],
        "atlas_structure_list": ["WHOLEHEART"],
        "atlas_path": ATLAS_PATH,
        "atlas_image_format": "Case_{0}/Images/Case_{0}_CROP.nii.gz",
        "atlas_label_format": "Case_{0}/Structures/Case_{0}_{1}_CROP.nii.gz",
        "crop_atlas_to_structures": False,
        "crop_atlas_expansion_mm": (20, 20, 40),
    },
    "auto_crop_target_image_settings": {"expansion_mm": [20, 20, 40]},
    "linear_registration_settings": {
        "reg_method": "affine",
        "shrink_factors": [16, 8, 4],
        "smooth_sigmas": [0, 0, 0],
        "sampling_rate": 0.75,
        "default_value": None,
        "number_of_iterations": 50,
        "metric": "mean_squares",
        "optimiser": "gradient_descent_line_search",
        "verbose": False,
    },
    "deformable_registration_settings": {
        "isotropic_resample": True,
        "resolution_staging": [6, 3, 1.5],
        "iteration_staging": [150, 125, 100],
        "smoothing_sigmas": [0, 0, 0],
        "ncores": 8,
        "default_value": None,
        "verbose": False,
    },
    "label_fusion_settings": {"vote_type": "unweighted", "vote_params": None, "optimal_threshold": {}},
    # iterative atlas removal, schema of cardiac/run.py:156-165; off unless a reference structure is named
    "iar_settings": {
        "reference_structure": False,
        "smooth_distance_maps": True,
        "smooth_sigma": 1,
        "z_score_statistic": "mad",
        "outlier_method": "iqr",
        "outlier_factor": 1.5,
        "min_best_atlases": 5,
        "project_on_sphere": False,
    },
    "postprocessing_settings": {
        "run_postprocessing": True,
        "binaryfillhole_mm": 3,
        "structures_for_binaryfillhole": [],
        "structures_for_overlap_correction": [],
    },
}

QUICK_REG_SETTINGS = {  # multiatlas/run.py:205-215
    "reg_method": "similarity",
    "shrink_factors": [8],
    "smooth_sigmas": [0],
    "sampling_rate": 0.75,
    "default_value": -1000,
    "number_of_iterations": 25,
    "final_interp": sitkLinear,
    "metric": "mean_squares",
    "optimiser": "gradient_descent_line_search",
}


class _Dist:
    """torch.distributed when initialised, a single-rank stand-in otherwise."""

    def __init__(self):
        import torch.distributed as dist

        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1

    def all_reduce_sum(self, t):
        if self.dist:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def all_gather(self, t):
        """-> list of world tensors shaped like t (rank order)."""
        if not self.dist:
            return [t]
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return out


_STREAM_POOL = {}   # device index -> long-lived worker streams (each owns one pp_ctx + workspace, see runtime.context)


def _worker_streams(device, n):
    pool = _STREAM_POOL.setdefault(device.index, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device))
    return pool[:n]


def _map_atlases(fn, ids, streams_per_gpu, device):
    """Run fn(atlas_id) for this rank's atlases, `streams_per_gpu` at a time.  Each worker thread runs under its own
    long-lived HIP stream, hence its own pp_ctx (runtime.context follows torch's current stream), so one atlas's
    small coarse-level kernels overlap another's; ctypes calls release the GIL."""
    if streams_per_gpu <= 1 or len(ids) <= 1 or device.type != "cuda":
        return {i: fn(i) for i in ids}
    import queue

    main = torch.cuda.current_stream(device)
    ready = torch.cuda.Event()
    ready.record(main)
    free = queue.SimpleQueue()
    for s in _worker_streams(device, min(streams_per_gpu, len(ids))):
        free.put(s)

    def work(i):
        torch.cuda.set_device(device)
        s = free.get()
        try:
            s.wait_event(ready)
            with torch.cuda.stream(s):
                out = fn(i)
                done = torch.cuda.Event()
                done.record(s)
            return out, done
        finally:
            free.put(s)

    with ThreadPoolExecutor(max_workers=min(streams_per_gpu, len(ids))) as ex:
        res = dict(zip(ids, ex.map(work, ids)))
    out = {}
    for i, (val, done) in res.items():
        main.wait_event(done)
        out[i] = val
    return out


def run_segmentation(img, settings=MUTLIATLAS_SETTINGS_DEFAULTS, atlases=None, streams_per_gpu=1, return_atlas_set=False):
    """Runs the atlas-based segmentation (reference multiatlas/run.py:106-441).

    img: target Image (replicated on every rank).  atlases: {atlas_id: {"CT Image": Image, <structure>: Image}}
    holding at least this rank's share (atlas_id_list[rank::world_size]); a rank may hold them all.
    Returns (results, results_prob): {structure: uint8 Image}, {structure: fp32 probability Image}, on every rank.
    """
    img = as_image(img)
    settings = copy.deepcopy(settings)
    dd = _Dist()
    device = img.device
    atlas_id_list = list(settings["atlas_settings"]["atlas_id_list"])
    atlas_structure_list = list(settings["atlas_settings"]["atlas_structure_list"])
    my_ids = atlas_id_list[dd.rank::dd.world]
    if atlases is None:     # multiatlas/run.py:155-170: read this rank's atlases from disk
        from ..io import read_image

        a = settings["atlas_settings"]
        atlases = {}
        for atlas_id in my_ids:
            entry = {"CT Image": read_image(f"{a['atlas_path']}/{a['atlas_image_format'].format(atlas_id)}", device)}
            for struct in atlas_structure_list:
                entry[struct] = read_image(f"{a['atlas_path']}/{a['atlas_label_format'].format(atlas_id, struct)}", device)
            atlases[atlas_id] = entry

    # ---- initialisation: optional crop of each atlas to its structures (:172-190) ----
    atlas_set = {}
    for atlas_id in my_ids:
        src = atlases[atlas_id]
        image = as_image(src["CT Image"])
        structures = {s: as_image(src[s]) for s in atlas_structure_list if s in src}
        if settings["atlas_settings"].get("crop_atlas_to_structures", False):
            size, index = label_to_roi(list(structures.values()), expansion_mm=settings["atlas_settings"]["crop_atlas_expansion_mm"])
            image = crop_to_roi(image, size, index)
            structures = {s: crop_to_roi(v, size, index) for s, v in structures.items()}
        atlas_set[atlas_id] = {"Original": {"CT Image": image, **structures}}

    # ---- step 1: automatic cropping of the target (:203-249) ----
    expansion_mm = settings["auto_crop_target_image_settings"]["expansion_mm"]
    crop_ids = atlas_id_list[: min(8, len(atlas_id_list))]
    mine = [a for a in crop_ids if a in atlas_set]

    def quick(atlas_id):
        reg_image, _ = linear_registration(img, atlas_set[atlas_id]["Original"]["CT Image"], **QUICK_REG_SETTINGS)
        return reg_image.tensor.float()

    acc = torch.zeros(img.shape, dtype=torch.float32, device=device)
    for t in _map_atlases(quick, mine, streams_per_gpu, device).values():
        acc += t
    dd.all_reduce_sum(acc)
    combined = img.like(((acc / float(len(crop_ids))) > -1000).to(torch.uint8))
    crop_box_size, crop_box_index = label_to_roi(combined, expansion_mm=expansion_mm)
    img_crop = crop_to_roi(img, crop_box_size, crop_box_index)
    del acc, combined

    # ---- steps 2-4a, per atlas: linear -> propagate -> demons -> propagate -> weight map (:261-362) ----
    lin_set = settings["linear_registration_settings"]
    dir_set = settings["deformable_registration_settings"]
    vote_type = settings["label_fusion_settings"]["vote_type"]
    vote_params = settings["label_fusion_settings"]["vote_params"]

    def chain(atlas_id):
        orig = atlas_set[atlas_id]["Original"]
        _, initial_tfm = linear_registration(img_crop, orig["CT Image"], **lin_set)
        rir = {"Transform": initial_tfm,
               "CT Image": apply_transform(orig["CT Image"], img_crop, initial_tfm, -1000, sitkLinear)}
        for s in atlas_structure_list:
            if s in orig:
                rir[s] = apply_transform(orig[s], img_crop, initial_tfm, 0, sitkNearestNeighbor)
        _, dir_tfm, _ = fast_symmetric_forces_demons_registration(img_crop, rir["CT Image"], **dir_set)
        out = {"Transform": dir_tfm,
               "CT Image": apply_transform(rir["CT Image"], transform=dir_tfm, default_value=-1000, interpolator=sitkLinear)}
        for s in atlas_structure_list:
            if s in rir:
                out[s] = apply_transform(rir[s], transform=dir_tfm, default_value=0, interpolator=sitkNearestNeighbor)
        out["Weight Map"] = compute_weight_map(img_crop, out["CT Image"], vote_type=vote_type, vote_params=vote_params)
        return out

    for atlas_id, out in _map_atlases(chain, my_ids, streams_per_gpu, device).items():
        atlas_set[atlas_id]["Original"] = None
        atlas_set[atlas_id]["DIR"] = out

    # ---- optional: iterative atlas removal (cardiac/run.py:879-891 -> label/iar.py) ----
    iar = dict(settings.get("iar_settings") or {})
    ref_struct = iar.pop("reference_structure", False)
    if ref_struct:
        from ..label.iar import run_iar

        # every rank needs every atlas's propagated reference structure + weight map: all_gather them slot by slot
        # (slot k of rank r is atlas_id_list[r + k * world]); all ranks then run the same, deterministic selection.
        slots = (len(atlas_id_list) + dd.world - 1) // dd.world
        full_set = {}
        for k in range(slots):
            have = k < len(my_ids)
            if have:
                d = atlas_set[my_ids[k]]["DIR"]
                m = (d[ref_struct].tensor != 0).to(torch.uint8).contiguous()
                w = d["Weight Map"].tensor.float().contiguous()
            else:
                m = torch.zeros(img_crop.shape, dtype=torch.uint8, device=device)
                w = torch.zeros(img_crop.shape, dtype=torch.float32, device=device)
            ms, ws = dd.all_gather(m), dd.all_gather(w)
            for r in range(dd.world):
                idx = r + k * dd.world
                if idx < len(atlas_id_list):
                    full_set[atlas_id_list[idx]] = {"DIR": {"Weight Map": img_crop.like(ws[r]), ref_struct: img_crop.like(ms[r])}}
        full_set = {i: full_set[i] for i in atlas_id_list}          # reference order
        kept = run_iar(atlas_set=full_set, reference_structure=ref_struct, **iar)
        removed = [i for i in atlas_id_list if i not in kept]
        if removed:
            logger.info("IAR removed atlases: %s", removed)
        my_ids = [i for i in my_ids if i in kept]
        run_segmentation.last_iar_removed = removed
        del full_set, kept

    # ---- step 4b: label fusion, the one cross-atlas exchange (fusion.py:263-288) ----
    ctx = runtime.context(device)
    S, n = len(atlas_structure_list), img_crop.tensor.numel()
    buf = torch.zeros((2 * S,) + img_crop.shape, dtype=torch.float32, device=device)   # [wsum_s, wlsum_s] per structure
    for atlas_id in my_ids:
        d = atlas_set[atlas_id]["DIR"]
        w = d["Weight Map"].tensor.contiguous()
        for k, s in enumerate(atlas_structure_list):
            if s in d:
                lab = d[s].tensor
                lab = (lab if lab.dtype == torch.uint8 else lab.to(torch.uint8)).contiguous()
                ctx.fuse_accumulate(w, lab, buf[2 * k], buf[2 * k + 1], n)
    dd.all_reduce_sum(buf)
    combined_label_dict = {s: finalize_probability(ctx, img_crop, buf[2 * k], buf[2 * k + 1]) for k, s in enumerate(atlas_structure_list)}
    del buf

    # ---- step 6: threshold, largest component, paste back into the target's space (:373-404) ----
    results, results_prob = {}, {}
    template_binary = img.like(torch.zeros(img.shape, dtype=torch.uint8, device=device))
    template_prob = img.like(torch.zeros(img.shape, dtype=torch.float32, device=device))
    for s in atlas_structure_list:
        prob = combined_label_dict[s]
        thr = settings["label_fusion_settings"]["optimal_threshold"].get(s, 0.5)
        binary = process_probability_image(prob, thr)
        results[s] = paste(template_binary, binary, crop_box_index)
        results_prob[s] = paste(template_prob, prob, crop_box_index)

    # ---- step 8: post-processing (:409-437) ----
    pp = settings["postprocessing_settings"]
    if pp["run_postprocessing"]:
        if pp["structures_for_binaryfillhole"]:
            from scipy import ndimage
            import numpy as np

            radius = [int(pp["binaryfillhole_mm"] / sp) for sp in img.GetSpacing()]
            for s in pp["structures_for_binaryfillhole"]:
                if s not in results:
                    continue
                a = results[s].numpy() > 0
                lab, ncomp = ndimage.label(a)
                if ncomp:
                    counts = np.bincount(lab.ravel())[1:]
                    a = lab == 1 + int(np.argmax(counts))           # RelabelComponent(...) == 1: the largest
                zz, yy, xx = np.ogrid[-radius[2]:radius[2] + 1, -radius[1]:radius[1] + 1, -radius[0]:radius[0] + 1]
                ball = ((xx / max(radius[0], 0.5)) ** 2 + (yy / max(radius[1], 0.5)) ** 2 + (zz / max(radius[2], 0.5)) ** 2) <= 1.0
                a = ndimage.binary_closing(np.pad(a, [(r, r) for r in radius[::-1]]), structure=ball)
                a = a[radius[2]:a.shape[0] - radius[2], radius[1]:a.shape[1] - radius[1], radius[0]:a.shape[2] - radius[0]]
                results[s] = img.like(torch.from_numpy(a.astype(np.uint8)).to(device))
        if len(pp["structures_for_overlap_correction"]) >= 2:      # :425-434
            from ..label.utils import correct_volume_overlap

            fixed = correct_volume_overlap({s: results[s] for s in pp["structures_for_overlap_correction"]})
            for s in pp["structures_for_overlap_correction"]:
                results[s] = fixed[s]

    if return_atlas_set:
        return results, results_prob, atlas_set
    return results, results_prob
