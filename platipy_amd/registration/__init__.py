from .deformable import fast_symmetric_forces_demons_registration, multiscale_demons, HipDemonsFilter  # noqa: F401
from .linear import linear_registration  # noqa: F401
from .utils import (  # noqa: F401
    apply_deformable_transform,
    apply_linear_transform,
    apply_transform,
    convert_mask_to_distance_map,
    convert_mask_to_reg_structure,
    smooth_and_resample,
)
