"""Drop-in for the pieces of platipy/imaging/label/utils.py the pipelines use after fusion:
correct_volume_overlap (:23-58).  Element-wise tensor arithmetic on the GPU."""
import numpy as np
import torch

from ..image import as_image


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
