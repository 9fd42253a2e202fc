"""Drop-in for platipy/imaging/generation/mask.py:107-159 (extend_mask), used by the structure-guided cardiac
pipeline (projects/cardiac/run.py:741-746, 826-831).  A handful of slice copies on the device tensor."""
import numpy as np
import torch

from ..image import as_image


def extend_mask(mask, direction=("ax", "sup"), extension_mm=10, interior_mm_shape=10):
    """Extend a binary label a number of slices along the axial direction; the extended part takes the shape of
    the union (max) of `interior_mm_shape` worth of the label's end slices.  Only the "ax" axis is handled, as in
    the reference ("PROTOTYPE!", mask.py:110).  Python slice semantics are kept as the reference has them,
    including the empty slices its negative/inverted ranges produce."""
    mask = as_image(mask)
    t = mask.tensor
    vals = torch.unique(t[t > 0])
    if len(vals) > 2:
        cutoff = float(np.median(vals.cpu().numpy()))
        arr = ((t >= cutoff) & (t <= float(vals.max()))).to(torch.uint8)
    else:
        arr = t.clone()
    if direction[0] == "ax":
        occupied = torch.nonzero((arr != 0).flatten(1).any(dim=1)).flatten()
        if occupied.numel() == 0:
            raise ValueError("extend_mask: the mask is empty")   # the reference fails in np.min of an empty array
        inferior_slice, superior_slice = int(occupied.min()), int(occupied.max())
        n_slices_ext = int(extension_mm / mask.GetSpacing()[2])
        n_slices_est = int(interior_mm_shape / mask.GetSpacing()[2])

        def shape_of(lo, hi):
            sl = arr[lo:hi]
            if sl.shape[0] == 0:
                raise ValueError("extend_mask: no interior slices to take the shape from")   # np.max over an empty axis
            return sl.max(dim=0).values

        if direction[1] == "sup":
            max_index = min(arr.shape[0], superior_slice + 1 + n_slices_ext)
            for s_in in range(superior_slice + 1 - n_slices_est, max_index):
                arr[s_in] = shape_of(superior_slice - n_slices_est, superior_slice)
        if direction[1] == "inf":
            min_index = max(arr.shape[0], inferior_slice - n_slices_ext + n_slices_est)
            for s_in in range(min_index, inferior_slice):
                arr[s_in] = shape_of(inferior_slice + n_slices_est, inferior_slice)
    return mask.like(arr)
