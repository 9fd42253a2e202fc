"""Drop-in for platipy/imaging/utils/crop.py:24-77 (label_to_roi, crop_to_roi) plus the paste-back the
pipelines do with sitk.Paste (multiatlas/run.py:386-404).  The bounding box is one streaming HIP pass
(pp_bounding_box); the rest is index arithmetic."""
import numpy as np
import torch

from ..image import Image, as_image


def label_to_roi(label, expansion_mm=[0, 0, 0], return_as_list=False):
    """Bounding box of a binary label (or the union of several), expanded by `expansion_mm` and clipped
    to the image: returns (crop_box_size, crop_box_index), both (x, y, z)."""
    from .. import runtime

    if isinstance(label, (list, tuple)) or (hasattr(label, "__iter__") and not isinstance(label, Image)):
        labels = [as_image(l) for l in label]
    else:
        labels = [as_image(label)]
    ref = labels[0]
    spacing = np.array(ref.GetSpacing())
    ctx = runtime.context(ref.device)
    lo, hi = [2 ** 31 - 1] * 3, [-1] * 3
    for l in labels:     # one streaming pass per label on the device (pp_bounding_box); boxes merged here
        t = l.tensor
        if t.dtype not in (torch.uint8, torch.float32):
            t = (t > 0).to(torch.uint8)
        box = ctx.bounding_box(t.contiguous(), l.GetSize(), t.dtype == torch.float32)
        if box[0] <= box[1]:
            lo = [min(lo[k], box[2 * k]) for k in range(3)]
            hi = [max(hi[k], box[2 * k + 1]) for k in range(3)]
    if hi[0] < 0:
        raise ValueError("label_to_roi: empty label")
    idx, size = lo, [hi[k] - lo[k] + 1 for k in range(3)]
    index, size = np.array(idx), np.array(size)
    expansion = (np.array(expansion_mm) / spacing).astype(int)
    crop_box_index = np.max([index - expansion, np.array([0, 0, 0])], axis=0)
    crop_box_size = np.min([np.array(ref.GetSize()) - crop_box_index, size + 2 * expansion], axis=0)
    crop_box_size = [int(i) for i in crop_box_size]
    crop_box_index = [int(i) for i in crop_box_index]
    if return_as_list:
        return crop_box_index + crop_box_size
    return crop_box_size, crop_box_index


def crop_to_roi(image, size, index):
    """sitk.RegionOfInterest: the sub-volume keeps its physical position (origin moves to the first kept voxel)."""
    image = as_image(image)
    x0, y0, z0 = (int(i) for i in index)
    sx, sy, sz = (int(s) for s in size)
    t = image.tensor[..., z0:z0 + sz, y0:y0 + sy, x0:x0 + sx].contiguous()
    d = np.asarray(image.direction).reshape(3, 3)
    origin = np.asarray(image.origin) + (d * np.asarray(image.spacing)[None, :]) @ np.array([x0, y0, z0], dtype=np.float64)
    return Image(t, image.spacing, tuple(origin), image.direction, image.is_vector)


def crop_to_label_extent(image, label, expansion_mm=0):
    if not hasattr(expansion_mm, "__iter__"):
        expansion_mm = [expansion_mm] * 3
    size, index = label_to_roi(label, expansion_mm=expansion_mm)
    return crop_to_roi(image, size, index)


def paste(template, source, index):
    """sitk.Paste(template, source, source.GetSize(), (0, 0, 0), index): copy `source` into a copy of `template`."""
    template, source = as_image(template), as_image(source)
    out = template.tensor.clone()
    x0, y0, z0 = (int(i) for i in index)
    sz, sy, sx = source.shape
    out[z0:z0 + sz, y0:y0 + sy, x0:x0 + sx] = source.tensor.to(out.dtype)
    return template.like(out)
