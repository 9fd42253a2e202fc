"""Drop-in for platipy/imaging/label/iar.py:59-301 (run_iar) and label/projection.py:67-92
(evaluate_distance_to_reference): iterative atlas removal.

GPU: consensus fusion (combine_labels), thresholding + largest component (process_probability_image), the
exact Euclidean distance map of every test contour and the reference's label contour.  Host: the statistics
on the sampled distances (<= 1e5 numbers per atlas) -- MAD z-scores, histogram, Gaussian fit, IQR fence --
exactly as the reference computes them with numpy/scipy.  The spherical-projection branch
(project_on_sphere=True; off by default, cardiac/run.py:163) is not implemented.
"""
import logging
import sys

import numpy as np
import torch
from scipy.optimize import curve_fit

from .. import runtime
from ..image import as_image
from .fusion import combine_labels, process_probability_image

logger = logging.getLogger(__name__)


def median_absolute_deviation(data, axis=None):
    return np.median(np.abs(data - np.median(data, axis=axis)), axis=axis)


_NORM_PDF_C = np.sqrt(2 * np.pi)


def gaussian_curve(x, a, m, s):
    """a * scipy.stats.norm.pdf(x, loc=m, scale=s) (reference iar.py:55-56), evaluated with scipy's own operations in scipy's
    order -- y = (x - m) / s, exp(-y^2 / 2) / sqrt(2 pi), / s, NaN for a scale that is not positive -- without the
    rv_continuous argument machinery around them (32 us a call, a hundred calls per curve_fit, one curve_fit per atlas and
    pass: 3 of the 3.7 ms of _q_metric).  Same bits: tests/test_iar.py holds the two equal."""
    x = np.asarray(x, dtype=np.float64)
    if not s > 0:
        return np.full(x.shape, np.nan) * a
    y = (x - m) / s
    return a * (np.exp(-y ** 2 / 2.0) / _NORM_PDF_C / s)


def distance_map(mask, signed=True, inside_positive=False):
    """sitk.SignedMaurerDistanceMap(mask, squaredDistance=False, useImageSpacing=True) (sitk.Abs of it when
    signed=False): fp32 Image of distances (mm) to the object's border voxels."""
    mask = as_image(mask)
    ctx = runtime.context(mask.device)
    m = (mask.tensor != 0).to(torch.uint8).contiguous()
    out = torch.empty(mask.shape, dtype=torch.float32, device=m.device)
    ctx.distance_map(m, mask.geom(), out, signed=signed, inside_positive=inside_positive)
    return mask.like(out)


def label_contour(mask):
    """sitk.LabelContour(mask): uint8 Image of the object's face-connected boundary voxels."""
    mask = as_image(mask)
    ctx = runtime.context(mask.device)
    m = (mask.tensor != 0).to(torch.uint8).contiguous()
    out = torch.empty_like(m)
    ctx.label_contour(m, mask.GetSize(), out)
    return mask.like(out)


def evaluate_distance_to_reference(reference_volume, test_volume, resample_factor=1):
    """Distance from every point of the reference surface to the test volume's surface (projection.py:67-92)."""
    test_distance_map = distance_map(test_volume, signed=False)
    ref_surface = label_contour(reference_volume).tensor == 1
    surface_values = test_distance_map.tensor[ref_surface]      # raster order, like numpy boolean indexing
    return surface_values[::resample_factor].cpu().numpy()


def _z_scores(own, others, statistic):
    """Robust z-score of one atlas's distance samples against the remaining atlases (iar.py:166-199), numpy form
    (kept as the readable statement of the rule; run_iar uses the device form below)."""
    others = np.asarray(others)
    kind = statistic.lower()
    if kind == "std":
        centre, spread = others.mean(axis=0), others.std(axis=0)
        if np.any(spread == 0):
            spread[spread == 0] = spread.mean()
    elif kind == "mad":
        centre = np.median(others, axis=0)
        spread = 1.4826 * median_absolute_deviation(others, axis=0)
        if np.any(spread == 0):
            spread[spread == 0] = np.median(spread)
    else:
        raise ValueError("z_score must be one of: MAD, STD")
    return np.ravel((own - centre) / spread)


def _median0(t):
    """np.median along axis 0 of a float32 [rows, n] tensor, numpy's rule: the middle row of the sorted column, or
    the float32 mean of the two middle rows."""
    rows = t.shape[0]
    srt = torch.sort(t, dim=0).values
    if rows % 2:
        return srt[rows // 2]
    return (srt[rows // 2 - 1] + srt[rows // 2]) / 2


def _z_scores_device(samples, k, statistic):
    """_z_scores(samples[k], the other rows, statistic) on the device the samples live on ([atlases, n] float32): the
    leave-one-out medians are A sorts of (A - 1) x n, seconds of numpy partitions at 32 atlases, milliseconds here."""
    own = samples[k]
    others = torch.cat([samples[:k], samples[k + 1:]], dim=0)
    kind = statistic.lower()
    if kind == "std":
        centre, spread = others.mean(dim=0), others.std(dim=0, unbiased=False)
        zero = spread == 0
        if bool(zero.any()):
            spread = torch.where(zero, spread.mean(), spread)
    elif kind == "mad":
        centre = _median0(others)
        spread = 1.4826 * _median0((others - centre).abs())
        zero = spread == 0
        if bool(zero.any()):
            s_sorted = torch.sort(spread).values       # np.median of the 1-D spread vector
            n = s_sorted.numel()
            med = s_sorted[n // 2] if n % 2 else (s_sorted[n // 2 - 1] + s_sorted[n // 2]) / 2
            spread = torch.where(zero, med, spread)
    else:
        raise ValueError("z_score must be one of: MAD, STD")
    return ((own - centre) / spread).flatten().cpu().numpy()


def _q_metric(z):
    """Excess of the z-score histogram over its best-fit Gaussian, weighted by z^2 (iar.py:211-230)."""
    density, edges = np.histogram(z, bins=np.linspace(-15, 15, 501), density=True)
    centres = (edges[1:] + edges[:-1]) / 2.0
    try:
        popt, _ = curve_fit(f=gaussian_curve, xdata=centres, ydata=density)
        ideal = gaussian_curve(centres, *popt)
    except (RuntimeError, ValueError):
        ideal = gaussian_curve(centres, a=1, m=density.mean(), s=density.std())
    trap = np.trapezoid if hasattr(np, "trapezoid") else np.trapz
    return np.float64(trap(np.abs(density - ideal) * np.abs(centres) ** 2, centres))


def _outlier_limit(q_values, method, factor, min_best):
    """Fence computed on all but (at most) the three worst atlases (iar.py:232-249)."""
    finite = [q for q in q_values if ~np.isnan(q) and np.isfinite(q)]
    best = np.sort(finite)[: max([min_best, len(finite) - 3])]
    kind = method.lower()
    if kind == "iqr":
        q75, q25 = np.percentile(best, [75, 25], axis=0)
        return q75 + factor * (q75 - q25)
    if kind == "std":
        return np.mean(best, axis=0) + factor * np.std(best, axis=0)
    logger.error(" outlier_method must be one of: IQR, STD")
    sys.exit()


def _support_roi(consensus, margin=2):
    """(size, index) of the box around the consensus probability's support, `margin` voxels wider and clipped to the image.
    Every atlas's label lies inside it (a voxel any atlas labels has a positive vote; the smoothing only widens the support),
    and so does every later pass's consensus (removing atlases removes votes).  Fill-hole, largest component, the distance
    map and the contour computed on this box equal the full-volume results at its voxels: everything outside it is
    background connected to the image border, and a distance is a distance to the label's border voxels, all inside.
    None: no support at all -- the caller works on the whole volume."""
    ctx = runtime.context(consensus.device)
    t = consensus.tensor if consensus.tensor.dtype == torch.float32 else consensus.tensor.float()
    box = ctx.bounding_box(t.contiguous(), consensus.GetSize(), True)
    if box[0] > box[1]:
        return None
    n = consensus.GetSize()
    lo = [max(0, box[2 * k] - margin) for k in range(3)]
    hi = [min(n[k] - 1, box[2 * k + 1] + margin) for k in range(3)]
    return [hi[k] - lo[k] + 1 for k in range(3)], lo


def _in_roi(image, roi):
    from ..utils.crop import crop_to_roi

    return as_image(image) if roi is None else crop_to_roi(as_image(image), roi[0], roi[1])


def _test_distance_map(cache, atlas_id, label_image, roi=None):
    """abs(SignedMaurerDistanceMap) of one atlas's propagated reference structure, flattened (projection.py:80-82 after
    iar.py's process_probability_image(..., 0.1)), on the box `roi` (_support_roi).  The reference recomputes it in every
    pass; it depends on the atlas alone -- only the consensus contour it is SAMPLED on changes between passes -- so it is
    computed once per atlas and kept for the run (fp32, one box per atlas)."""
    if atlas_id not in cache:
        cache[atlas_id] = distance_map(process_probability_image(_in_roi(label_image, roi), 0.1), signed=False).tensor.flatten()
    return cache[atlas_id]


def run_iar(atlas_set, reference_structure, smooth_distance_maps=False, smooth_sigma=1, z_score_statistic="MAD",
            outlier_method="IQR", min_best_atlases=10, outlier_factor=1.5, iteration=0, single_step=False,
            project_on_sphere=False, label="DIR", _distance_maps=None, _roi=False):
    """Perform iterative atlas removal on the atlas_set (reference iar.py:59-301): build the consensus contour,
    sample every atlas's distance to it, score each atlas by how non-Gaussian its robust z-scores are, drop the
    ones beyond the fence, repeat until nothing is dropped."""
    if project_on_sphere:
        raise NotImplementedError("project_on_sphere=True is outside this build's scope (SURVEY 2)")
    ids = list(atlas_set.keys())
    consensus = combine_labels(atlas_set, reference_structure, label=label)[reference_structure]
    # iar.py:104-112: fewer atlases -> thinner sampling (the "< 7" branch there is unreachable and stays so here)
    resample_factor = 5 if len(ids) < 12 else 1
    # the structure occupies a small part of the volume (config 5's heart: 4.8 of 67 Mvoxel): everything below works on the box
    # around the FIRST pass's consensus support, which holds every label and every later consensus (_support_roi)
    roi = _support_roi(consensus) if _roi is False else _roi
    reference_volume = process_probability_image(_in_roi(consensus, roi), threshold=0.95)
    # evaluate_distance_to_reference for every atlas (projection.py:67-92): the reference contour is the same for all of
    # them, so its voxel list is built once; the samples stay on the device for the leave-one-out statistics
    ref_index = torch.nonzero((label_contour(reference_volume).tensor == 1).flatten()).flatten()[::resample_factor]
    cache = {} if _distance_maps is None else _distance_maps
    samples = torch.stack([_test_distance_map(cache, i, atlas_set[i][label][reference_structure], roi)[ref_index] for i in ids])

    q_results = {}
    for k, atlas_id in enumerate(ids):
        q_results[atlas_id] = _q_metric(_z_scores_device(samples, k, z_score_statistic))
    limit = _outlier_limit(list(q_results.values()), outlier_method, outlier_factor, min_best_atlases)
    keep = [i for i, q in q_results.items() if q <= limit]
    run_iar.last_q_results = dict(q_results)       # diagnostic hook for tests / logging
    logger.info("IAR step %d: limit %.4g, Q = %s", iteration, limit, {i: round(float(q), 4) for i, q in q_results.items()})

    if len(keep) == len(ids):
        return atlas_set
    survivors = {i: atlas_set[i] for i in keep}
    if single_step:
        return survivors
    return run_iar(atlas_set=survivors, reference_structure=reference_structure, smooth_distance_maps=smooth_distance_maps,
                   smooth_sigma=smooth_sigma, z_score_statistic=z_score_statistic, outlier_method=outlier_method,
                   min_best_atlases=min_best_atlases, outlier_factor=outlier_factor, iteration=iteration + 1,
                   project_on_sphere=project_on_sphere, label=label, _distance_maps={i: cache[i] for i in keep}, _roi=roi)


def run_iar_distributed(dd, atlas_set, my_ids, atlas_id_list, reference_structure, target, weights, smooth_distance_maps=False,
                        smooth_sigma=1, z_score_statistic="MAD", outlier_method="IQR", min_best_atlases=10, outlier_factor=1.5,
                        project_on_sphere=False, label="DIR"):
    """run_iar when the atlases are spread over ranks (one process per GPU): the same passes, with each rank doing the
    work of its own atlases only and three small exchanges per pass --
      * the consensus: all_reduce(SUM) of [sum w, sum w L] for the reference structure (w = the atlas's global-vote
        weight, one number), finalised identically everywhere;
      * the distance samples on the consensus contour: all_gather of [slots, n] float32 per rank (n ~ 1e4-1e5);
      * the Q values: all_reduce(SUM) of a vector with each rank's own entries filled in.
    Every rank takes the same keep / drop decision from the same numbers.  `dd`: world, rank, all_reduce_sum(t),
    all_gather(t); `weights`: {atlas_id: float} for this rank's atlases; `target`: the (cropped) target Image.
    Returns the list of kept atlas ids (reference order)."""
    if project_on_sphere:
        raise NotImplementedError("project_on_sphere=True is outside this build's scope (SURVEY 2)")
    from .. import runtime
    from .fusion import finalize_probability, label_tensor

    device = target.device
    ctx = runtime.context(device)
    kept = list(atlas_id_list)
    slots = (len(atlas_id_list) + dd.world - 1) // dd.world
    iteration = 0
    cache = {}      # this rank's atlases' distance maps: computed once, sampled on every pass's consensus contour
    roi = False     # the box around the first pass's consensus support (_support_roi): the same on every rank
    while True:
        mine = [i for i in my_ids if i in kept]
        buf = torch.zeros((2,) + tuple(target.shape), dtype=torch.float32, device=device)
        for i in mine:
            lab = label_tensor(atlas_set[i][label][reference_structure]).to(torch.float32)   # the rule combine_labels applies
            buf[0] += float(weights[i])
            buf[1] += float(weights[i]) * lab
        dd.all_reduce_sum(buf)
        consensus = finalize_probability(ctx, target, buf[0].contiguous(), buf[1].contiguous())
        resample_factor = 5 if len(kept) < 12 else 1
        if roi is False:
            roi = _support_roi(consensus)
        reference_volume = process_probability_image(_in_roi(consensus, roi), threshold=0.95)
        ref_index = torch.nonzero((label_contour(reference_volume).tensor == 1).flatten()).flatten()[::resample_factor]
        n = int(ref_index.numel())
        local = torch.zeros((slots, n), dtype=torch.float32, device=device)
        for k, i in enumerate(my_ids):
            if i in kept:
                local[k] = _test_distance_map(cache, i, atlas_set[i][label][reference_structure], roi)[ref_index]
            else:
                cache.pop(i, None)
        gathered = dd.all_gather(local)
        rows = [gathered[idx % dd.world][idx // dd.world].to(device) for idx, aid in enumerate(atlas_id_list) if aid in kept]
        samples = torch.stack(rows)
        q = torch.zeros(len(kept), dtype=torch.float64, device=device)
        for pos, aid in enumerate(kept):
            if aid in mine:
                q[pos] = float(_q_metric(_z_scores_device(samples, pos, z_score_statistic)))
        dd.all_reduce_sum(q)
        q_results = dict(zip(kept, q.cpu().numpy()))
        limit = _outlier_limit(list(q_results.values()), outlier_method, outlier_factor, min_best_atlases)
        keep = [i for i in kept if q_results[i] <= limit]
        run_iar_distributed.last_q_results = dict(q_results)
        logger.info("IAR step %d (distributed): limit %.4g, Q = %s", iteration, limit, {i: round(float(v), 4) for i, v in q_results.items()})
        if len(keep) == len(kept):
            return kept
        kept = keep
        iteration += 1
