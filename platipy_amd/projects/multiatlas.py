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
from ..image import as_image
from ..label.fusion import compute_weight_map, finalize_probability, label_tensor, process_probability_image
from ..registration.deformable import fast_symmetric_forces_demons_registration
from ..registration.linear import linear_registration
from ..generation.mask import extend_mask
from ..registration.utils import apply_transform, convert_mask_to_reg_structure
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
    """torch.distributed when initialised, a single-rank stand-in otherwise.  Every exchange is timed under a label
    (`timings`: label -> milliseconds, summed): CUDA events on the current stream for device tensors -- no host
    synchronisation is added, the events are read when `timings_ms()` is asked for -- perf_counter for host tensors (gloo)."""

    def __init__(self):
        import torch.distributed as dist

        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1
        self._events, self._host_ms, self.label = [], {}, "other"

    def _timed(self, t, fn):
        if not self.dist:
            return fn()
        if t.is_cuda:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn()
            b.record()
            self._events.append((self.label, a, b))
            return out
        import time

        t0 = time.perf_counter()
        out = fn()
        self._host_ms[self.label] = self._host_ms.get(self.label, 0.0) + 1e3 * (time.perf_counter() - t0)
        return out

    def timings_ms(self):
        """label -> total milliseconds spent in that label's exchanges on this rank (synchronises the recorded events)."""
        out = dict(self._host_ms)
        for label, a, b in self._events:
            b.synchronize()
            out[label] = out.get(label, 0.0) + a.elapsed_time(b)
        return out

    def all_reduce_sum(self, t):
        if self.dist:
            self._timed(t, lambda: self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM))
        return t

    def reduce_sum_to_root(self, t):
        """SUM onto rank 0 (the `RCCL reduce` of north_star): half the ring traffic of an all_reduce; other ranks' buffers
        hold partial sums afterwards and must not be used."""
        if self.dist:
            self._timed(t, lambda: self.dist.reduce(t, dst=0, op=self.dist.ReduceOp.SUM))
        return t

    def all_reduce_min(self, t):
        if self.dist:
            self._timed(t, lambda: self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN))
        return t

    def all_gather(self, t):
        """-> list of world tensors shaped like t (rank order)."""
        if not self.dist:
            return [t]
        out = [torch.empty_like(t) for _ in range(self.world)]
        self._timed(t, lambda: self.dist.all_gather(out, t))
        return out


_STREAM_POOL = {}   # device index -> long-lived worker streams (each owns one pp_ctx + workspace, see runtime.context)


def _worker_streams(device, n):
    pool = _STREAM_POOL.setdefault(device.index, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device))
    return pool[:n]


def _hand_over(obj, stream, _seen=None):
    """Tell the caching allocator that every tensor reachable from `obj` (dicts, sequences, Images, transforms and the fields
    inside them) is now used on `stream` too (torch.Tensor.record_stream): a block a worker stream allocated is then not
    handed out again -- to the worker's next atlas -- while work queued on the consumer's stream still reads it.  The
    `done` -> `main.wait_event` ordering below already makes today's hand-over correct; this keeps it correct for whatever a
    caller does with the results afterwards, whichever stream frees them."""
    _seen = set() if _seen is None else _seen
    if obj is None or id(obj) in _seen:
        return
    _seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _hand_over(v, stream, _seen)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _hand_over(v, stream, _seen)
    else:
        for name in ("tensor", "field", "transforms"):     # Image / DisplacementFieldTransform / CompositeTransform
            if hasattr(obj, name):
                _hand_over(getattr(obj, name), stream, _seen)


# How chains that share a GPU are scheduled against each other (runtime.Turnstile).  STAGGER: one chain at a time through its
# throughput-bound phase (finest demons level + the full-resolution resamples behind it), the others' latency-bound phases
# underneath.  ENTRY_SLOTS: how many chains may be in their linear stage at once (0: no bound) -- chains started together are
# thereby admitted one after the other and reach the turnstile one throughput-bound phase apart instead of all at once.
STAGGER = False
ENTRY_SLOTS = 0


def _map_atlases(fn, ids, streams_per_gpu, device):
    """Run fn(atlas_id) for this rank's atlases, `streams_per_gpu` at a time.  Each worker thread runs under its own
    long-lived HIP stream, hence its own pp_ctx (runtime.context follows torch's current stream), so one atlas's
    small coarse-level kernels overlap another's; ctypes calls release the GIL.  The chains are staggered (STAGGER,
    ENTRY_SLOTS above): results are those of the sequential run bit for bit, only the order of the device work changes."""
    if streams_per_gpu <= 1 or len(ids) <= 1 or device.type != "cuda":
        return {i: fn(i) for i in ids}
    import queue
    import threading

    turnstile = runtime.Turnstile() if STAGGER else None
    slots = threading.Semaphore(ENTRY_SLOTS) if (STAGGER and ENTRY_SLOTS > 0) else None

    main = torch.cuda.current_stream(device)
    ready = torch.cuda.Event()
    ready.record(main)
    free = queue.SimpleQueue()
    for s in _worker_streams(device, min(streams_per_gpu, len(ids))):
        free.put(s)

    def work(i):
        torch.cuda.set_device(device)
        s = free.get()
        runtime.set_turnstile(turnstile, slots)
        try:
            s.wait_event(ready)
            with torch.cuda.stream(s):
                out = fn(i)
                done = torch.cuda.Event()
                done.record(s)
            return out, done
        finally:
            runtime.set_turnstile(None)
            free.put(s)

    with ThreadPoolExecutor(max_workers=min(streams_per_gpu, len(ids))) as ex:
        res = dict(zip(ids, ex.map(work, ids)))
    out = {}
    for i, (val, done) in res.items():
        main.wait_event(done)
        _hand_over(val, main)
        out[i] = val
    return out


def _mask_outside(image, mask, outside_value):
    """sitk.Mask(image, mask, outsideValue): image where mask != 0, outsideValue elsewhere."""
    t = image.tensor
    return image.like(torch.where(mask != 0, t, torch.full((), outside_value, dtype=t.dtype, device=t.device)))


def run_segmentation(img, settings=MUTLIATLAS_SETTINGS_DEFAULTS, atlases=None, streams_per_gpu=1, return_atlas_set=False,
                     fusion_collective="all_reduce"):
    """Runs the atlas-based segmentation (reference multiatlas/run.py:106-441).

    img: target Image (replicated on every rank).  atlases: {atlas_id: {"CT Image": Image, <structure>: Image}}
    holding at least this rank's share (atlas_id_list[rank::world_size]); a rank may hold them all.
    Returns (results, results_prob): {structure: uint8 Image}, {structure: fp32 probability Image}.
    fusion_collective: "all_reduce" (default) -- every rank receives the sums, finalises identically and returns the
    results; "reduce" -- the sums go to rank 0 only (an RCCL reduce: half the xGMI bytes), rank 0 alone finalises and
    returns the results, the other ranks return empty dicts (the reference's single caller is rank 0).
    """
    out = atlas_pipeline(img, settings, None, atlases, streams_per_gpu, cardiac=False, fusion_collective=fusion_collective)
    run_segmentation.last_iar_removed = out["iar_removed"]
    run_segmentation.last_fusion_payload_bytes = out["fusion_payload_bytes"]
    run_segmentation.last_exchange_ms = out["exchange_ms"]
    run_segmentation.last_world_size = out["world_size"]
    run_segmentation.last_crop_box = out["crop_box"]        # (size, index), both (x, y, z): multiatlas/run.py:241-243
    if return_atlas_set:
        return out["results"], out["results_prob"], out["atlas_set"]
    return out["results"], out["results_prob"]


def atlas_pipeline(img, settings, guide_structure=None, atlases=None, streams_per_gpu=1, cardiac=False, fusion_collective="all_reduce"):
    """The skeleton shared by run_segmentation (multiatlas/run.py:106-441) and run_cardiac_segmentation
    (cardiac/run.py:507-1147): read atlases -> crop the target -> per atlas [linear -> (structure-guided demons)
    -> demons -> propagate] -> (iterative atlas removal) -> weight maps -> fuse -> threshold -> paste back ->
    post-process.  `cardiac` selects the cardiac pipeline's result conventions (only structures with an
    optimal_threshold are voted, the guide structure is handed back, return_as_cropped).
    Returns a dict: results, results_prob, atlas_set, iar_removed, img_crop."""
    if fusion_collective not in ("all_reduce", "reduce"):
        raise ValueError("fusion_collective must be 'all_reduce' or 'reduce'")
    img = as_image(img)
    settings = copy.deepcopy(settings)
    dd = _Dist()
    device = img.device
    a_set = settings["atlas_settings"]
    atlas_id_list = list(a_set["atlas_id_list"])
    atlas_structure_list = list(a_set["atlas_structure_list"])
    guided = guide_structure is not None     # the reference tests `if guide_structure:` on a sitk.Image (always true)
    guide_structure_name = a_set.get("guide_structure_name") if guided else None
    if guided:
        guide_structure = as_image(guide_structure)
        if not guide_structure_name:
            raise KeyError("atlas_settings['guide_structure_name'] is needed with a guide structure")
    my_ids = atlas_id_list[dd.rank::dd.world]
    if atlases is None:     # multiatlas/run.py:155-170, cardiac/run.py:560-568: read this rank's atlases from disk
        from ..io import read_image

        atlases = {}
        for atlas_id in my_ids:
            entry = {"CT Image": read_image(f"{a_set['atlas_path']}/{a_set['atlas_image_format'].format(atlas_id)}", device)}
            for struct in atlas_structure_list:
                entry[struct] = read_image(f"{a_set['atlas_path']}/{a_set['atlas_label_format'].format(atlas_id, struct)}", device)
            atlases[atlas_id] = entry

    # ---- initialisation: optional crop of each atlas to its structures (:172-190) ----
    atlas_set = {}
    for atlas_id in my_ids:
        src = atlases[atlas_id]
        image = as_image(src["CT Image"])
        structures = {s: as_image(src[s]) for s in atlas_structure_list if s in src}
        if a_set.get("crop_atlas_to_structures", False):
            size, index = label_to_roi(list(structures.values()), expansion_mm=a_set["crop_atlas_expansion_mm"])
            image = crop_to_roi(image, size, index)
            structures = {s: crop_to_roi(v, size, index) for s, v in structures.items()}
        atlas_set[atlas_id] = {"Original": {"CT Image": image, **structures}}

    # ---- step 1: automatic cropping of the target (multiatlas :203-249; cardiac :603-659) ----
    expansion_mm = settings["auto_crop_target_image_settings"]["expansion_mm"]
    if guided:
        crop_box_size, crop_box_index = label_to_roi(guide_structure, expansion_mm=expansion_mm)
        img_crop = crop_to_roi(img, crop_box_size, crop_box_index)
        guide_structure = crop_to_roi(guide_structure, crop_box_size, crop_box_index)
        target_reg_structure = convert_mask_to_reg_structure(guide_structure, expansion=2)
        superior_extension = a_set["superior_extension"]
        expanded_target_mask = extend_mask(guide_structure, direction=("ax", "sup"), extension_mm=superior_extension,
                                           interior_mm_shape=superior_extension / 2)
    else:
        crop_ids = atlas_id_list[: min(8, len(atlas_id_list))]
        mine = [a for a in crop_ids if a in atlas_set]

        def quick(atlas_id):
            reg_image, _ = linear_registration(img, atlas_set[atlas_id]["Original"]["CT Image"], **QUICK_REG_SETTINGS)
            return reg_image.tensor.float()

        acc = torch.zeros(img.shape, dtype=torch.float32, device=device)
        for t in _map_atlases(quick, mine, streams_per_gpu, device).values():
            acc += t
        dd.label = "crop_allreduce"
        dd.all_reduce_sum(acc)
        combined = img.like(((acc / float(len(crop_ids))) > -1000).to(torch.uint8))
        crop_box_size, crop_box_index = label_to_roi(combined, expansion_mm=expansion_mm)
        img_crop = crop_to_roi(img, crop_box_size, crop_box_index)
        del acc, combined

    # ---- steps 2-3, per atlas: linear -> propagate -> [guided demons -> propagate] -> demons -> propagate ----
    lin_set = settings["linear_registration_settings"]
    dir_set = settings["deformable_registration_settings"]
    sg_set = settings.get("structure_guided_registration_settings") if guided else None

    def chain(atlas_id):
        orig = atlas_set[atlas_id]["Original"]
        if guided:      # cardiac :682-688: register the guide structures' distance-map images, not the CTs
            target_reg_image = target_reg_structure
            atlas_reg_image = convert_mask_to_reg_structure(orig[guide_structure_name], expansion=2)
        else:
            target_reg_image, atlas_reg_image = img_crop, orig["CT Image"]
        with runtime.entry_slot():
            _, initial_tfm = linear_registration(target_reg_image, atlas_reg_image, **lin_set)
        cur = {"Transform": initial_tfm,
               "CT Image": apply_transform(orig["CT Image"], img_crop, initial_tfm, -1000, sitkLinear)}
        for s in atlas_structure_list:
            if s in orig:
                cur[s] = apply_transform(orig[s], img_crop, initial_tfm, 0, sitkNearestNeighbor)
        if guided:      # cardiac :700-722, :766-816
            reg_mask = apply_transform(atlas_reg_image, img_crop, initial_tfm, 0, sitkLinear)
            expanded = extend_mask(orig[guide_structure_name], direction=("ax", "sup"), extension_mm=superior_extension,
                                   interior_mm_shape=superior_extension / 2)
            expanded = apply_transform(expanded, img_crop, initial_tfm, 0, sitkNearestNeighbor)
            _, sg_tfm, _ = fast_symmetric_forces_demons_registration(target_reg_structure, reg_mask, **sg_set)
            nxt = {"Transform": sg_tfm,
                   "CT Image": apply_transform(cur["CT Image"], transform=sg_tfm, default_value=-1000, interpolator=sitkLinear)}
            expanded = apply_transform(expanded, img_crop, sg_tfm, 0, sitkNearestNeighbor)
            for s in atlas_structure_list:
                if s in cur:
                    nxt[s] = apply_transform(cur[s], transform=sg_tfm, default_value=0, interpolator=sitkNearestNeighbor)
            cur = nxt
            # cardiac :834-849: both images are masked to the union of the extended guide structures and to the
            # atlas image's soft tissue before the intensity-driven demons
            combined_mask = torch.maximum(expanded.tensor, expanded_target_mask.tensor.to(expanded.tensor.dtype))
            atlas_reg_image = _mask_outside(cur["CT Image"], combined_mask, -1000)
            atlas_reg_image = _mask_outside(atlas_reg_image, atlas_reg_image.tensor > -400, -1000)
            target_reg_image = _mask_outside(img_crop, combined_mask, -1000)
            target_reg_image = _mask_outside(target_reg_image, atlas_reg_image.tensor > -400, -1000)
        else:
            target_reg_image, atlas_reg_image = img_crop, cur["CT Image"]
        _, dir_tfm, _ = fast_symmetric_forces_demons_registration(target_reg_image, atlas_reg_image, **dir_set)
        # ("Linear Transform": the stage-2 result, which the reference drops with its "RIR" entry at run.py:347 -- kept for
        # diagnostics and the whole-pipeline parity test; no function of the path reads it)
        out = {"Transform": dir_tfm, "Linear Transform": initial_tfm,
               "CT Image": apply_transform(cur["CT Image"], transform=dir_tfm, default_value=-1000, interpolator=sitkLinear)}
        for s in atlas_structure_list:
            if s in cur:
                out[s] = apply_transform(cur[s], transform=dir_tfm, default_value=0, interpolator=sitkNearestNeighbor)
        return out

    for atlas_id, out in _map_atlases(chain, my_ids, streams_per_gpu, device).items():
        atlas_set[atlas_id]["Original"] = None
        atlas_set[atlas_id]["DIR"] = out

    # ---- step 4: iterative atlas removal on global-vote weights (cardiac/run.py:879-891 -> label/iar.py) ----
    iar = dict(settings.get("iar_settings") or {})
    ref_struct = iar.pop("reference_structure", False)
    removed = []
    if ref_struct:
        from ..label.iar import run_iar, run_iar_distributed

        dd.label = "iar_exchange"
        if dd.dist is not None:   # (also a process group of ONE rank: the exchanges then run through its communicator)
            # every rank scores its own atlases; consensus, distance samples and Q values are exchanged (label/iar.py)
            weights = {i: float(compute_weight_map(img_crop, atlas_set[i]["DIR"]["CT Image"], vote_type="global").tensor.flatten()[0])
                       for i in my_ids}
            kept = run_iar_distributed(dd, {i: atlas_set[i] for i in my_ids}, my_ids, atlas_id_list, ref_struct, img_crop, weights, **iar)
        else:
            full_set = {}
            for i in atlas_id_list:
                d = atlas_set[i]["DIR"]
                full_set[i] = {"DIR": {"Weight Map": compute_weight_map(img_crop, d["CT Image"], vote_type="global"), ref_struct: d[ref_struct]}}
            kept = list(run_iar(atlas_set=full_set, reference_structure=ref_struct, **iar))
            del full_set
        removed = [i for i in atlas_id_list if i not in kept]
        if removed:
            logger.info("IAR removed atlases: %s", removed)
        my_ids = [i for i in my_ids if i in kept]
    else:
        logger.info("IAR: No reference structure, skipping iterative atlas removal.")

    # ---- step 5: weight maps + label fusion, the one cross-atlas exchange (fusion.py:263-288) ----
    vote_type = settings["label_fusion_settings"]["vote_type"]
    vote_params = settings["label_fusion_settings"]["vote_params"]
    ctx = runtime.context(device)
    S, n = len(atlas_structure_list), img_crop.tensor.numel()
    # Payload of the one data-path collective.  When every atlas (on every rank) carries every structure the weight sum is
    # the same for all structures, so the buffer is [sum w, sum w L_1 .. sum w L_S] = (1 + S) volumes; an atlas
    # that lacks a structure does not vote on it (fusion.py:263-276), which needs a weight sum per structure: 2 S volumes.
    dd.label = "fusion_layout"
    complete = torch.tensor([1 if all(s in atlas_set[a]["DIR"] for a in my_ids for s in atlas_structure_list) else 0],
                            dtype=torch.int32, device=device if dd.dist is None or dd.dist.get_backend() == "nccl" else "cpu")
    shared_wsum = bool(int(dd.all_reduce_min(complete).item()))
    buf = torch.zeros(((1 + S) if shared_wsum else 2 * S,) + img_crop.shape, dtype=torch.float32, device=device)
    for atlas_id in my_ids:
        d = atlas_set[atlas_id]["DIR"]
        d["Weight Map"] = compute_weight_map(img_crop, d["CT Image"], vote_type=vote_type, vote_params=vote_params)
        w = d["Weight Map"].tensor.contiguous()
        for k, s in enumerate(atlas_structure_list):
            if shared_wsum:
                ctx.fuse_accumulate(w, label_tensor(d[s]), buf[0] if k == 0 else None, buf[1 + k], n)
            elif s in d:
                ctx.fuse_accumulate(w, label_tensor(d[s]), buf[2 * k], buf[2 * k + 1], n)
    dd.label = "fusion_allreduce" if fusion_collective == "all_reduce" else "fusion_reduce"
    if fusion_collective == "all_reduce":
        dd.all_reduce_sum(buf)
    else:
        dd.reduce_sum_to_root(buf)
    fusion_payload_bytes = buf.numel() * 4
    dd.label = "other"
    root_only = fusion_collective == "reduce" and dd.world > 1
    if root_only and dd.rank != 0:      # the sums live on rank 0: nothing to finalise here
        del buf
        return {"results": {}, "results_prob": {}, "atlas_set": atlas_set, "iar_removed": removed, "img_crop": img_crop,
                "crop_box": (list(crop_box_size), list(crop_box_index)),
                "fusion_payload_bytes": fusion_payload_bytes, "exchange_ms": dd.timings_ms(), "world_size": dd.world}
    combined_label_dict = {s: finalize_probability(ctx, img_crop, buf[0] if shared_wsum else buf[2 * k], buf[1 + k] if shared_wsum else buf[2 * k + 1])
                           for k, s in enumerate(atlas_structure_list)}
    del buf

    # ---- step 6: threshold, largest component, paste back into the target's space (:373-404; cardiac :928-1004) ----
    results, results_prob = {}, {}
    thresholds = settings["label_fusion_settings"]["optimal_threshold"]
    as_cropped = bool(cardiac and settings.get("return_as_cropped", False))
    template_binary = img.like(torch.zeros(img.shape, dtype=torch.uint8, device=device))
    template_prob = img.like(torch.zeros(img.shape, dtype=torch.float32, device=device))
    # the cardiac pipeline votes only the structures it has a threshold for (:931-932)
    vote_structures = [s for s in thresholds if s in atlas_structure_list] if cardiac else atlas_structure_list
    for s in vote_structures:
        prob = combined_label_dict[s]
        binary = process_probability_image(prob, thresholds.get(s, 0.5))
        if cardiac and settings.get("return_proba_as_contours", False):      # cardiac/run.py:945-951, 964-970
            # atlas k's contour in bit k + 1 (label/utils.py:219-254).  Each rank encodes its own atlases at their positions
            # in the global list; the bit sets are disjoint, so one integer all_reduce(SUM) assembles the image everywhere
            enc = torch.zeros(img_crop.shape, dtype=torch.int64, device=device)
            for pos, a in enumerate(atlas_id_list):
                if a in atlas_set and "DIR" in atlas_set[a] and s in atlas_set[a]["DIR"]:
                    enc |= (process_probability_image(atlas_set[a]["DIR"][s], 0.5).tensor != 0).to(torch.int64) << (pos + 1)
            if len(atlas_id_list) > 32:
                raise ValueError("You can only encode a maximum of 32 structures with this method!")
            if root_only:
                raise NotImplementedError("return_proba_as_contours gathers every rank's contours: use fusion_collective='all_reduce'")
            dd.label = "contour_allreduce"
            dd.all_reduce_sum(enc)
            prob = img_crop.like(enc)
            template_p = img.like(torch.zeros(img.shape, dtype=prob.tensor.dtype, device=device))
        else:
            template_p = template_prob
        if as_cropped:
            results[s], results_prob[s] = binary, prob
        else:
            results[s] = paste(template_binary, binary, crop_box_index)
            results_prob[s] = paste(template_p, prob, crop_box_index)
        if cardiac and guided and not settings.get("return_atlas_guide_structure", False):   # :955-960, :994-1004
            g = guide_structure.like(guide_structure.tensor.to(torch.uint8))
            g = g if as_cropped else paste(template_binary, g, crop_box_index)
            results[guide_structure_name] = g
            results_prob[guide_structure_name] = g

    # ---- step 8: post-processing (:409-437; cardiac :1113-1141), on the device ----
    pp = settings["postprocessing_settings"]
    if pp["run_postprocessing"]:
        from ..label.utils import binary_morphological_closing, correct_volume_overlap, largest_component

        radius = [int(pp["binaryfillhole_mm"] / sp) for sp in img.GetSpacing()]
        for s in pp["structures_for_binaryfillhole"]:
            if s not in results:
                continue
            results[s] = binary_morphological_closing(largest_component(results[s]), radius)
        overlap = pp["structures_for_overlap_correction"]
        if cardiac or len(overlap) >= 2:      # the multi-atlas pipeline skips fewer than two (:425); cardiac does not
            fixed = correct_volume_overlap({s: results[s] for s in overlap})
            for s in overlap:
                results[s] = fixed[s]

    if as_cropped:
        results["CROP_IMAGE"] = img_crop
    return {"results": results, "results_prob": results_prob, "atlas_set": atlas_set, "iar_removed": removed, "img_crop": img_crop,
            "crop_box": (list(crop_box_size), list(crop_box_index)),
            "fusion_payload_bytes": fusion_payload_bytes, "exchange_ms": dd.timings_ms(), "world_size": dd.world}
