"""Drop-in for the pieces of platipy/imaging/label/utils.py the pipelines use after fusion:
correct_volume_overlap (:23-58, element-wise tensor arithmetic on the GPU), plus the binary-mask SimpleITK calls
the pipelines make inline -- BinaryDilate / BinaryErode / BinaryMorphologicalClosing with the ball kernel and
"RelabelComponent(ConnectedComponent(x)) == 1" -- as HIP kernels (pp_morph.hip, pp_cc.hip)."""
import numpy as np
import torch

from .. import runtime
from ..image import as_image

_DILATE, _ERODE, _CLOSE = 0, 1, 2


def _u8(image):
    t = image.tensor
    return (t if t.dtype == torch.uint8 else (t != 0).to(torch.uint8)).contiguous()


def _radius3(image, radius):
    if not hasattr(radius, "__iter__"):
        radius = [radius] * 3
    radius = [int(r) for r in radius]
    if len(radius) != 3 or min(radius) < 0:
        raise ValueError(f"kernel radius must be three non-negative voxel counts, got {radius}")
    return radius


def _morph(mask, radius, op):
    mask = as_image(mask)
    radius = _radius3(mask, radius)
    src = _u8(mask)
    out = torch.empty_like(src)
    runtime.context(mask.device).binary_morph_ball(src, mask.GetSize(), radius, op, out)
    return mask.like(out)


def binary_dilate(mask, radius):
    """sitk.BinaryDilate(mask, radius): ball kernel, radius in voxels (x, y, z) (registration/utils.py:328-329)."""
    return _morph(mask, radius, _DILATE)


def binary_erode(mask, radius):
    """sitk.BinaryErode(mask, radius): ball kernel, the image boundary counts as foreground."""
    return _morph(mask, radius, _ERODE)


def binary_morphological_closing(mask, radius):
    """sitk.BinaryMorphologicalClosing(mask, radius), safe border (multiatlas/run.py:422, cardiac/run.py:1128)."""
    return _morph(mask, radius, _CLOSE)


def largest_component(mask):
    """sitk.RelabelComponent(sitk.ConnectedComponent(mask)) == 1 (multiatlas/run.py:421): the largest
    face-connected component, the first in raster order on ties; an empty mask stays empty."""
    mask = as_image(mask)
    src = _u8(mask)
    out = torch.empty_like(src)
    runtime.context(mask.device).fillhole_largest_component(src, mask.GetSize(), out, fill_holes=False)
    return mask.like(out)


def correct_volume_overlap(binary_label_dict, assign_overlap_to_largest=True):
    """Make the structures disjoint: rank them by volume (largest first by default) and give every voxel to the
    first structure in that order that contains it.  The reference does this by prime-encoding the labels
    (label/utils.py:44-56); the result is the same set arithmetic."""
    names = list(binary_label_dict.keys())
    labels = {k: as_image(v) for k, v in binary_label_dict.items()}
    vals = [int((labels[k].tensor != 0).sum()) for k in names]
    rank = np.argsort(vals)[::-1] if assign_overlap_to_largest else np.argsort(vals)
    ranked = [names[i] for i in rank]
    first = labels[ranked[0]]
    taken = torch.zeros(first.shape, dtype=torch.bool, device=first.device)
    out = {}
    for k in ranked:
        m = (labels[k].tensor != 0) & ~taken
        taken |= m
        out[k] = labels[k].like(m.to(torch.uint8))
    return out


def binary_encode_structure_list(structure_list):
    """Encode up to 32 binary labels into one integer image, structure k in bit k + 1 (reference label/utils.py:219-254).
    The reference casts to UInt32, which cannot hold bit 32; the tensor here is int64."""
    if len(structure_list) > 32:
        raise ValueError("You can only encode a maximum of 32 structures with this method!")
    first = as_image(structure_list[0])
    enc = torch.zeros(first.shape, dtype=torch.int64, device=first.device)
    for power, s_img in enumerate(structure_list):
        enc |= (as_image(s_img).tensor != 0).to(torch.int64) << (power + 1)
    return first.like(enc)


def binary_decode_image(binary_encoded_img):
    """Decode a binary-encoded label map into the list of non-empty structures (reference label/utils.py:257-288)."""
    img = as_image(binary_encoded_img)
    enc = img.tensor.to(torch.int64)
    out = []
    for power in range(32):
        s = (enc & (1 << (power + 1))) != 0
        if bool(s.any()):
            out.append(img.like(s.to(torch.uint8)))
    return out
