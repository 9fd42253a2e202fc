"""Drop-in for platipy/imaging/label/fusion.py: compute_weight_map (:56-202), combine_labels (:239-292)
and process_probability_image (:295-328) on volumes resident in HBM.

Out of scope, as in SURVEY 8(a8): vote_type="patch_correlation" (a Python per-patch loop in the
reference, :119-128), combine_labels_staple and mutual_information raise NotImplementedError.
"""
import numpy as np
import torch

from .. import runtime
from ..image import Image, as_image

DEFAULT_VOTE_PARAMS = {
    "sigma": 2.0,
    "epsilon": 1e-5,
    "factor": 1e12,
    "gain": 6,
    "blockSize": 5,
    "normalise": False,
}


def _f32(image):
    t = image.tensor
    return (t if t.dtype == torch.float32 else t.float()).contiguous()


def compute_weight_map(target_image, moving_image, vote_type="unweighted", vote_params=None):
    """Computes the weight map (reference fusion.py:56-202).  fp32 result on the target grid."""
    target_image, moving_image = as_image(target_image), as_image(moving_image)
    p = dict(DEFAULT_VOTE_PARAMS)
    if vote_params:
        p.update(vote_params)
    ctx = runtime.context(target_image.device)
    t, m = _f32(target_image), _f32(moving_image)   # :77-80 (quirk N1: everything in float32)
    n = t.numel()
    vt = vote_type.lower()

    if vt == "patch_correlation":
        raise NotImplementedError("vote_type='patch_correlation' is outside the MI355X hot path (SURVEY 8 a8)")
    if vt == "unweighted":
        weight = t * 0.0 + 1.0                                                      # :151-152
    elif vt == "global":
        ssd = ctx.sum_sq_diff(t, m, n)                                              # :154-161, fp64 sum (quirk N5)
        weight = t * 0.0 + (p["factor"] / ssd)
    elif vt == "local":
        weight = torch.empty_like(t)
        ctx.weight_map_local(t, m, target_image.GetSize(), target_image.spacing, p["sigma"], p["epsilon"], weight)  # :163-169
        weight = _normalise(weight, p["normalise"])                                 # :171-177
    elif vt == "block":
        bs = p["blockSize"]
        bs = (bs,) * 3 if isinstance(bs, int) else tuple(bs)                         # (x, y, z) radii of sitk.BoxMean
        weight = torch.empty_like(t)
        ctx.weight_map_block(t, m, target_image.GetSize(), bs, p["factor"], p["gain"], weight)   # :179-190
        weight = _normalise(weight, p["normalise"])
    else:
        raise ValueError(f"unknown vote_type {vote_type!r}")
    return target_image.like(weight.float().contiguous())


def _normalise(weight, normalise):
    if isinstance(normalise, bool):
        if normalise:
            weight = weight / weight.max()
    elif isinstance(normalise, Image):
        mask = normalise.tensor != 0
        weight = weight / weight[mask].max()
    return weight


def label_tensor(image):
    """An atlas label as the fusion kernels take it: uint8 / bool masks stay uint8 (one byte per voxel through HBM);
    anything else is weighted as sitk.Cast(label, sitkFloat32) weights it (reference fusion.py:269-272) -- a
    probabilistic label of 0.7 contributes 0.7 w, an int16 label of 256 contributes 256 w."""
    t = as_image(image).tensor
    if t.dtype == torch.bool:
        t = t.to(torch.uint8)
    elif t.dtype != torch.uint8:
        t = t.to(torch.float32)
    return t.contiguous()


def _accumulate(ctx, atlas_set, case_ids, label, s_name):
    """Left fold over the atlases, as functools.reduce over sitk images does (:263, :269-276)."""
    first = atlas_set[case_ids[0]][label]["Weight Map"]
    wsum = torch.zeros(first.shape, dtype=torch.float32, device=first.device)
    wlsum = torch.zeros_like(wsum)
    n = wsum.numel()
    for cid in case_ids:
        w = _f32(as_image(atlas_set[cid][label]["Weight Map"]))
        ctx.fuse_accumulate(w, label_tensor(atlas_set[cid][label][s_name]), wsum, wlsum, n)
    return first, wsum, wlsum


def finalize_probability(ctx, ref, wsum, wlsum, threshold=1e-4, smooth_sigma=1.0):
    """P = wlsum / guarded(wsum) -> DiscreteGaussian(sigma^2) -> RescaleIntensity(0,1) -> Threshold (:264-288)."""
    n = wsum.numel()
    prob = torch.empty_like(wsum)
    ctx.fuse_divide(wlsum, wsum, prob, n)
    var = smooth_sigma * smooth_sigma
    ctx.discrete_gaussian(prob, prob, ref.GetSize(), ref.spacing, (var, var, var), 0.01, 32, True)
    lo, hi = ctx.minmax(prob, n)
    # sitk.Threshold(lower=threshold, upper=1, outsideValue=0); a falsy threshold skips it (lower = -inf here)
    ctx.rescale_threshold(prob, n, lo, hi, threshold if threshold else -3.0e38)
    return ref.like(prob)


def combine_labels(atlas_set, structure_name, label="DIR", threshold=1e-4, smooth_sigma=1.0):
    """Combine labels using weight maps (reference fusion.py:239-292).
    atlas_set[case_id][label] is a dict holding "Weight Map" and one Image per structure."""
    case_id_list = list(atlas_set.keys())
    if isinstance(structure_name, str):
        structure_name_list = [structure_name]
    elif isinstance(structure_name, list):
        structure_name_list = structure_name
    else:
        raise TypeError("structure_name must be a str or a list of str")
    combined_label_dict = {}
    for s_name in structure_name_list:
        valid = [i for i in case_id_list if s_name in atlas_set[i][label].keys()]
        if not valid:
            raise KeyError(f"no atlas holds structure {s_name!r}")
        ref = as_image(atlas_set[valid[0]][label]["Weight Map"])
        ctx = runtime.context(ref.device)
        ref, wsum, wlsum = _accumulate(ctx, atlas_set, valid, label, s_name)
        combined_label_dict[s_name] = finalize_probability(ctx, as_image(ref), wsum, wlsum, threshold, smooth_sigma)
    return combined_label_dict


def process_probability_image(probability_image, threshold=0.5):
    """Generate a mask given a probability image (reference fusion.py:295-328): /max, BinaryThreshold(>= thr),
    BinaryFillhole, ConnectedComponent, keep the largest component, uint8 -- all on the GPU (union-find labelling,
    pp_fillhole_largest_component_u8)."""
    if not isinstance(probability_image, Image):
        probability_image = Image(torch.as_tensor(np.asarray(probability_image)).to(runtime.default_device()))
    ctx = runtime.context(probability_image.device)
    prob = _f32(probability_image)
    n = prob.numel()
    if threshold > 0 and n >= CROP_MIN_VOXELS:
        # A fused probability is zero outside the (smoothed) union of the atlas labels -- a few per cent of a 512 x 512 x 256
        # volume -- and every voxel that can pass a POSITIVE threshold lies in the box around its support.  On that box (one
        # voxel wider, so that its rim is background connected to the outside) the threshold, the fill-hole, the labelling and
        # the largest component are those of the whole volume: everything outside is background reaching the image border.
        from ..utils.crop import crop_to_roi

        size = probability_image.GetSize()
        box = ctx.bounding_box(prob, size, True)
        if box[0] <= box[1]:
            lo = [max(0, box[2 * k] - 1) for k in range(3)]
            ext = [min(size[k] - 1, box[2 * k + 1] + 1) - lo[k] + 1 for k in range(3)]
            if ext[0] * ext[1] * ext[2] <= n // 2:
                inner = _process_probability_image(crop_to_roi(probability_image.like(prob), ext, lo), threshold)
                out = torch.zeros(prob.shape, dtype=torch.uint8, device=prob.device)
                out[lo[2]:lo[2] + ext[2], lo[1]:lo[1] + ext[1], lo[0]:lo[0] + ext[0]] = inner.tensor
                return probability_image.like(out)
    return _process_probability_image(probability_image.like(prob), threshold)


CROP_MIN_VOXELS = 1 << 22


def _process_probability_image(probability_image, threshold):
    ctx = runtime.context(probability_image.device)
    prob = _f32(probability_image)
    n = prob.numel()
    _, hi = ctx.minmax(prob, n)
    binary = torch.empty(prob.shape, dtype=torch.uint8, device=prob.device)
    ctx.binary_threshold(prob, n, hi, threshold, binary)
    out = torch.empty_like(binary)
    ctx.fillhole_largest_component(binary, probability_image.GetSize(), out, fill_holes=True)
    return probability_image.like(out)


def combine_labels_staple(label_list_dict, threshold=1e-4):
    raise NotImplementedError("STAPLE is outside the MI355X hot path (SURVEY 8 a8)")


def mutual_information(arr_a, arr_b, bins=64):
    raise NotImplementedError("mutual_information is outside the MI355X hot path (SURVEY 8 a8)")
