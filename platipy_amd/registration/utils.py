"""Drop-in for platipy/imaging/registration/utils.py:54-267 (apply_transform, its two wrappers and
smooth_and_resample), on torch tensors in HBM through the HIP C ABI."""
import functools
import threading
import logging

import numpy as np
import torch

from .. import _lib
from ..image import Image, as_image, cast_tensor
from .. import runtime
from ..transform import (
    CompositeTransform,
    DisplacementFieldTransform,
    Transform,
    sitkBSpline,
    sitkLinear,
    sitkNearestNeighbor,
)

logger = logging.getLogger(__name__)


def _check_interp(interpolator):
    """sitkNearestNeighbor / sitkLinear / sitkBSpline (cubic: itk::BSplineInterpolateImageFunction, spline order 3)."""
    if interpolator not in (sitkNearestNeighbor, sitkLinear, sitkBSpline):
        raise ValueError(f"unknown interpolator {interpolator!r}")
    return {sitkNearestNeighbor: _lib.INTERP_NEAREST, sitkLinear: _lib.INTERP_LINEAR, sitkBSpline: _lib.INTERP_BSPLINE}[interpolator]


def _split_transform(transform, reference):
    """-> (A, t, field_tensor_on_reference_grid or None) such that q = A p + t + field(p)."""
    if transform is None or type(transform) is Transform:
        return None, None, None
    parts = transform.flatten() if isinstance(transform, CompositeTransform) else [transform]
    parts = [p for p in parts if type(p) is not Transform]
    fields = [i for i, p in enumerate(parts) if isinstance(p, DisplacementFieldTransform)]
    if not fields:
        A, off = CompositeTransform(parts).matrix_offset()
        return A, off, None
    if len(fields) > 1 or fields[0] != len(parts) - 1:
        # any sitk.Transform may reach apply_transform (reference registration/utils.py:176-190): fields anywhere in the
        # composite, several of them, each on its own grid -> one total displacement field on the reference grid
        return None, None, _total_field(parts, reference)
    dvf = parts[-1].field
    if not dvf.same_grid(reference):
        dvf = resample_field(dvf, reference)
    f = dvf.tensor if dvf.tensor.dtype == torch.float32 else dvf.tensor.float()
    if len(parts) == 1:
        return None, None, f.contiguous()
    A, off = CompositeTransform(parts[:-1]).matrix_offset()
    # q = A (p + D(p)) + off = A p + off + A D(p)
    At = torch.tensor(A, dtype=torch.float32, device=f.device)
    f = torch.einsum("rc,czyx->rzyx", At, f).contiguous()
    return A, off, f


def _total_field(parts, reference):
    """D(p) = T(p) - p on `reference`'s grid for T = parts[0] o parts[1] o ... (the LAST member is applied first), members
    linear or displacement-field transforms in any order.  A field member F maps q -> q + F(q), with F interpolated
    linearly on ITS grid and zero outside it (itk::DisplacementFieldTransform); the running map is kept as p + D(p)."""
    ctx = runtime.context(reference.device)
    geom = reference.geom()
    A, off, D = np.eye(3), np.zeros(3), None        # pending linear map q = A p + off while no field has been applied yet
    for part in reversed(parts):
        if not isinstance(part, DisplacementFieldTransform):
            a, o = part.matrix_offset()
            if D is None:
                A, off = a @ A, a @ off + o
            else:   # q' = a (p + D) + o  ->  D' = (a - I) p + o + a D
                aD = torch.einsum("rc,czyx->rzyx", torch.tensor(a, dtype=torch.float32, device=D.device), D).contiguous()
                out = torch.empty_like(D)
                ctx.transform_to_field(geom, a, o, aD, out)
                D = out
            continue
        if D is None:
            D = torch.empty((3,) + reference.shape, dtype=torch.float32, device=reference.device)
            ctx.transform_to_field(geom, A, off, None, D)        # (A - I) p + off; zeros for the identity
        F = part.field
        Ft = (F.tensor if F.tensor.dtype == torch.float32 else F.tensor.float()).contiguous()
        if F.same_grid(reference) and _identity_direction(reference):
            ctx.compose_field(D, Ft, geom)                       # D(p) += F(p + D(p))
        else:   # the member's own grid -- or an oblique / flipped one, which pp_compose_field_f32 does not take (ADVICE round 3:
            # the same composite must not work or fail depending on whether a member happens to share the grid) --: each
            # component sampled at p + D(p) through the general resampler, linear, 0 outside
            add = torch.empty_like(D)
            for c in range(3):
                ctx.resample(Ft[c].contiguous(), F.geom(), geom, add[c], field=D, interp=_lib.INTERP_LINEAR, default_value=0.0, u8=False)
            D = D + add
    return D


def _identity_direction(image):
    return np.allclose(np.asarray(image.GetDirection(), dtype=np.float64).reshape(3, 3), np.eye(3), rtol=0.0, atol=1e-12)


def transform_to_displacement_field(transform, reference):
    """sitk.TransformToDisplacementField(transform, sitkVectorFloat64, reference grid) (reference deformable.py:101-108):
    D(p) = T(p) - p as a planar fp32 vector Image on `reference`'s grid.  Linear transforms, a displacement-field
    transform, or a composite of linear members whose last-listed (first-applied) member is a displacement field."""
    reference = as_image(reference)
    ctx = runtime.context(reference.device)
    A, off, field = _split_transform(transform, reference)
    out = torch.empty((3,) + reference.shape, dtype=torch.float32, device=reference.device)
    if A is None and field is None:
        out.zero_()
    elif A is None:
        out.copy_(field)
    else:
        ctx.transform_to_field(reference.geom(), A, off, field, out)
    return Image(out, reference.spacing, reference.origin, reference.direction, True)


def resample_field(field_image, reference, copy=True):
    """sitk.Resample(vector_image, reference): linear, identity transform, default 0.  `copy=False`: the caller owns
    `field_image` and lets the result alias it when no resampling is needed."""
    if (field_image.same_grid(reference) and field_image.tensor.dtype == torch.float32 and field_image.tensor.is_contiguous()):
        # the same grid (the finest level's up-sampling and deformable.py:185's final resample): linear interpolation at the
        # grid points returns the samples; a copy keeps the reference's "new image" semantics (callers update fields in place)
        t = field_image.tensor.clone() if copy else field_image.tensor
        return Image(t, reference.spacing, reference.origin, reference.direction, True)
    ctx = runtime.context(field_image.device)
    src = field_image.tensor if field_image.tensor.dtype == torch.float32 else field_image.tensor.float()
    out = torch.empty((3,) + reference.shape, dtype=torch.float32, device=src.device)
    ctx.resample_field(src.contiguous(), field_image.geom(), reference.geom(), out)
    return Image(out, reference.spacing, reference.origin, reference.direction, True)


def resample_image(image, reference, transform=None, interpolator=sitkLinear, default_value=0.0):
    """sitk.Resample(image, reference, transform, interpolator, default): result keeps fp32/uint8 storage."""
    interp = _check_interp(interpolator)
    ctx = runtime.context(image.device)
    A, t, field = _split_transform(transform, reference)
    if interp == _lib.INTERP_BSPLINE:
        # coefficients first (BSplineDecompositionImageFilter), then the 4 x 4 x 4 evaluation; integer images go through fp32
        src = (image.tensor if image.tensor.dtype == torch.float32 else image.tensor.float()).contiguous()
        coef = torch.empty_like(src)
        ctx.bspline_prefilter(src, image.GetSize(), coef)
        out = torch.empty(reference.shape, dtype=torch.float32, device=src.device)
        ctx.resample(coef, image.geom(), reference.geom(), out, affine_A=None if A is None else A.ravel(),
                     affine_t=None if A is None else t, field=field, interp=interp, default_value=float(default_value), u8=False)
        if image.tensor.dtype == torch.uint8:
            out = out.clamp_(0, 255).to(torch.uint8)
        return Image(out, reference.spacing, reference.origin, reference.direction, False)
    if image.tensor.dtype == torch.uint8:
        src, u8 = image.tensor, True
        out = torch.empty(reference.shape, dtype=torch.uint8, device=src.device)
    else:
        src, u8 = (image.tensor if image.tensor.dtype == torch.float32 else image.tensor.float()), False
        out = torch.empty(reference.shape, dtype=torch.float32, device=src.device)
    ctx.resample(src.contiguous(), image.geom(), reference.geom(), out, affine_A=None if A is None else A.ravel(),
                 affine_t=None if A is None else t, field=field, interp=interp, default_value=float(default_value), u8=u8)
    return Image(out, reference.spacing, reference.origin, reference.direction, False)


def apply_transform(input_image, reference_image=None, transform=None, default_value=0, interpolator=sitkNearestNeighbor):
    """Transform a volume or structure (reference: registration/utils.py:148-192)."""
    input_image = as_image(input_image)
    reference = as_image(reference_image) if reference_image is not None else input_image
    original_dtype = input_image.tensor.dtype
    out = resample_image(input_image, reference, transform, interpolator, default_value)
    return out.like(cast_tensor(out.tensor, original_dtype))


def apply_linear_transform(input_image, reference_image, transform, is_structure=False, default_value=0,
                           interpolator=sitkNearestNeighbor):
    """registration/utils.py:54-99"""
    if is_structure:
        if default_value != 0 or interpolator != sitkNearestNeighbor:
            logger.warning("is_structure is set to True, but you have set default_value and/or interpolator. "
                           "default_value and/or interpolator will be overwritten.")
        default_value = 0
        interpolator = sitkNearestNeighbor
    return apply_transform(input_image=input_image, reference_image=reference_image, transform=transform,
                           default_value=default_value, interpolator=interpolator)


def apply_deformable_transform(input_image, transform, is_structure=False, default_value=0, interpolator=sitkNearestNeighbor):
    """registration/utils.py:102-145"""
    if is_structure:
        if default_value != 0 or interpolator != sitkNearestNeighbor:
            logger.warning("is_structure is set to True, but you have set default_value and/or interpolator. "
                           "default_value and/or interpolator will be overwritten.")
        default_value = 0
        interpolator = sitkNearestNeighbor
    return apply_transform(input_image=input_image, reference_image=None, transform=transform, default_value=default_value,
                           interpolator=interpolator)


def discrete_gaussian(image, variance, maximum_kernel_width=32, maximum_error=0.01, use_image_spacing=True):
    """sitk.DiscreteGaussian on a scalar image (fp32 result)."""
    ctx = runtime.context(image.device)
    src = image.tensor if image.tensor.dtype == torch.float32 else image.tensor.float()
    var = [float(v) for v in np.broadcast_to(np.asarray(variance, dtype=np.float64), (3,))]
    out = torch.empty_like(src)
    ctx.discrete_gaussian(src.contiguous(), out, image.GetSize(), image.spacing, var, maximum_error, int(maximum_kernel_width),
                          use_image_spacing)
    return image.like(out)


def _rows_read_by_resample(n_in, n_out, ratio):
    """Input indices along one axis that a linear / nearest resample onto `n_out` corner-aligned samples touches
    (output j sits at input index j * ratio): floor and floor + 1, one more on either side where j * ratio is
    within 1e-6 of an integer (the kernel's own fp64 coordinate may land on either side of it)."""
    c = np.arange(n_out, dtype=np.float64) * ratio
    f = np.floor(c)
    frac = c - f
    idx = [f, f + 1, np.where(frac < 1e-6, f - 1, f), np.where(frac > 1 - 1e-6, f + 2, f + 1)]
    need = np.zeros(n_in, dtype=np.uint8)
    need[np.clip(np.concatenate(idx), 0, n_in - 1).astype(np.int64)] = 1
    return need


_NEED_MASKS = {}            # (sizes_in, sizes_out, ratios, device) -> masks; entries are NEVER evicted (see _need_masks)
_NEED_MASKS_LOCK = threading.Lock()
_NEED_MASKS_MAX = 256


def _need_masks(sizes_in, sizes_out, ratios, device):
    """(need_y, need_z) as uint8 device tensors for a pyramid level's geometry, or None when hardly anything can be skipped.
    Cached: a pipeline registers many pairs on the same grids, and each upload is a host-device round trip in front of
    the level's first kernel.  A cached tensor is read by kernels on whatever stream its caller runs on, so it must never go
    back to the allocator while such a kernel may be queued: entries are not evicted (a few hundred bytes each; past
    _NEED_MASKS_MAX geometries new ones are simply not cached) and the dictionary is dropped only by runtime.release_all(),
    which synchronises the device first (an LRU eviction could free a mask under another stream's kernel)."""
    key = (sizes_in, sizes_out, ratios, device)
    hit = _NEED_MASKS.get(key, False)
    if hit is not False:
        return hit
    need = [_rows_read_by_resample(n_in, n_out, ratio) for n_in, n_out, ratio in zip(sizes_in, sizes_out, ratios)]
    masks = None
    if need[0].mean() * need[1].mean() <= 0.5:
        dev = torch.device(device)
        masks = tuple(torch.from_numpy(n).to(dev) for n in need)
        if dev.type == "cuda":
            # the upload is ordered on THIS thread's stream only; the cached tensors are handed to every worker thread (each on
            # its own stream) from now on, so they must be complete before the cache publishes them
            torch.cuda.current_stream(dev).synchronize()
    with _NEED_MASKS_LOCK:
        if len(_NEED_MASKS) < _NEED_MASKS_MAX:
            _NEED_MASKS.setdefault(key, masks)
            return _NEED_MASKS[key]
    if masks is not None and torch.device(device).type == "cuda":
        for m in masks:     # uncached: the caller's stream is the only user, tell the allocator so
            m.record_stream(torch.cuda.current_stream(torch.device(device)))
    return masks


def release_cached_masks():
    """Called by runtime.release_all() after a device synchronisation."""
    with _NEED_MASKS_LOCK:
        _NEED_MASKS.clear()


def smooth_and_resample(image, isotropic_voxel_size_mm=None, shrink_factor=None, smoothing_sigma=None,
                        interpolator=sitkLinear, _share_input=False):
    """One pyramid level (reference: registration/utils.py:195-267): optional Gaussian blur with sigma in mm,
    then linear resampling onto a corner-aligned coarser grid.

    The result never shares storage with `image` (sitk.Resample returns a new image).  `_share_input=True` is the pyramid
    builder's private contract: it only READS its levels, so an unsmoothed shrink-factor-1 level may be the input's own
    tensor under a new header instead of a 268 MB copy.

    The blur is only evaluated where the resample will read it (pp_discrete_gaussian_rows_f32): onto an 8x coarser
    grid that is a quarter of the z planes and a sixteenth of the rows, and every value produced is the dense filter's."""
    image = as_image(image)
    as_image_tensor = image.tensor           # (the caller's storage: a blur below replaces `image` by a fresh tensor)
    original_spacing = image.GetSpacing()
    original_size = image.GetSize()

    if shrink_factor and isotropic_voxel_size_mm:
        raise AttributeError("Function must be called with either isotropic_voxel_size_mm or shrink_factor, not both.")
    elif isotropic_voxel_size_mm:
        scale_factor = isotropic_voxel_size_mm * np.ones(3) / np.array(image.GetSpacing())
        new_size = [int(sz / float(sf) + 0.5) for sz, sf in zip(original_size, scale_factor)]
    elif shrink_factor:
        if isinstance(shrink_factor, list):
            new_size = [int(sz / float(sf) + 0.5) for sz, sf in zip(original_size, shrink_factor)]
        else:
            new_size = [int(sz / float(shrink_factor) + 0.5) for sz in original_size]
    else:
        new_size = None

    new_spacing = None
    if new_size is not None:
        new_spacing = [((size_o_i - 1) * spacing_o_i) / (size_n_i - 1)
                       for size_o_i, spacing_o_i, size_n_i in zip(original_size, original_spacing, new_size)]

    if smoothing_sigma:
        if hasattr(smoothing_sigma, "__iter__"):
            smoothing_variance = [i * i for i in smoothing_sigma]
        else:
            smoothing_variance = (smoothing_sigma ** 2,) * 3
        maximum_kernel_width = int(max([8 * j * i for i, j in zip(image.GetSpacing(), smoothing_variance)]))
        need = None
        if new_size is not None and interpolator in (sitkLinear, sitkNearestNeighbor):
            need = _need_masks(tuple(original_size[1:]), tuple(new_size[1:]),
                               tuple(new_spacing[a] / original_spacing[a] for a in (1, 2)), str(image.device))
        if need is None:
            image = discrete_gaussian(image, smoothing_variance, maximum_kernel_width)
        else:
            ctx = runtime.context(image.device)
            src = image.tensor if image.tensor.dtype == torch.float32 else image.tensor.float()
            out = torch.empty_like(src)
            var = [float(v) for v in smoothing_variance]
            ctx.discrete_gaussian_rows(src.contiguous(), out, image.GetSize(), image.spacing, var, need[0], need[1], 0.01,
                                       int(maximum_kernel_width), True)
            image = image.like(out)      # valid exactly where the resample below reads it

    if new_size is None:
        # (no resampling asked for: the reference returns its -- possibly smoothed -- input image object itself)
        return image
    if (list(new_size) == list(original_size) and interpolator in (sitkLinear, sitkNearestNeighbor)
            and np.allclose(new_spacing, original_spacing, rtol=1e-12, atol=0.0)):
        # shrink factor 1 (the finest pyramid level): the output grid IS the input grid -- ((n - 1) s) / (n - 1) can differ from
        # s in the last bit, a shift of < 1e-9 voxel -- and resampling onto it returns the samples themselves
        shared = image.tensor if (_share_input or image.tensor is not as_image_tensor) else image.tensor.clone()
        return Image(shared, new_spacing, image.GetOrigin(), image.GetDirection())
    ref = Image(torch.empty((new_size[2], new_size[1], new_size[0]), dtype=torch.float32, device="meta"), new_spacing,
                image.GetOrigin(), image.GetDirection())
    out = resample_image(image, ref, None, interpolator, 0.0)
    return out.like(cast_tensor(out.tensor, image.tensor.dtype))


def convert_mask_to_distance_map(mask, squared_distance=False, normalise=False):
    """Generate a distance map from a binary label (reference registration/utils.py:270-299):
    sitk.SignedMaurerDistanceMap(mask, insideIsPositive=True, squaredDistance=..., useImageSpacing=True)."""
    from ..label.iar import distance_map

    mask = as_image(mask)
    t = mask.tensor
    vals = torch.unique(t[t > 0])
    if len(vals) > 2:   # more than one value: threshold at the median (:283-287)
        cutoff = float(np.median(vals.cpu().numpy()))
        mask = mask.like(((t >= cutoff) & (t <= float(vals.max()))).to(torch.uint8))
    raw = distance_map(mask, signed=True, inside_positive=True)
    if squared_distance:
        raw = raw.like(torch.sign(raw.tensor) * raw.tensor * raw.tensor)
    if normalise:
        return raw.like(raw.tensor / raw.tensor.max())
    return raw


def convert_mask_to_reg_structure(mask, expansion=(0, 0, 0), scale=lambda x: x):
    """Mask-like image for structure-guided registration (reference registration/utils.py:302-344): optional
    ball dilation (`expansion`: mm when scalar, voxels per axis when a sequence, as the reference has it), then
    the inside distance map, zero outside, scaled to [0, 1], float64."""
    from ..label.utils import binary_dilate

    mask = as_image(mask)
    t = mask.tensor
    vals = torch.unique(t[t > 0])
    if len(vals) > 2:   # more than one value: threshold at the median (:320-324)
        cutoff = float(np.median(vals.cpu().numpy()))
        mask = mask.like(((t >= cutoff) & (t <= float(vals.max()))).to(torch.uint8))
    if not hasattr(expansion, "__iter__"):
        expansion = [int(expansion / i) for i in mask.GetSpacing()]
    if any(expansion):
        mask = binary_dilate(mask, expansion)
    dm = convert_mask_to_distance_map(mask, squared_distance=False)
    inside = dm.tensor.double() * (mask.tensor != 0).to(torch.float64)
    return scale(dm.like(inside / inside.max()))
