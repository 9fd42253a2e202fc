"""platipy_amd -- MI355X-native deformable registration and multi-atlas label fusion.

Drop-in for the hot path of pyplati/platipy: `linear_registration`,
`fast_symmetric_forces_demons_registration`, `apply_transform`, `smooth_and_resample` and the
`label.fusion` utilities keep the reference's signatures; volumes live in HBM as torch tensors and
every voxel-level operation runs in hand-written HIP kernels behind the C ABI of
include/platipy_amd.h.  There is no CPU fallback.
"""
from .image import Image, image_from_array, array_from_image  # noqa: F401
from .transform import (  # noqa: F401
    AffineTransform,
    CompositeTransform,
    DisplacementFieldTransform,
    Euler3DTransform,
    FullAffineTransform,
    ScaleTransform,
    Similarity3DTransform,
    Transform,
    TranslationTransform,
    VersorRigid3DTransform,
    sitkBSpline,
    sitkLinear,
    sitkNearestNeighbor,
)

__version__ = "0.1.0"

from . import registration  # noqa: E402,F401
from . import label  # noqa: E402,F401
from . import generation  # noqa: E402,F401
from . import projects  # noqa: E402,F401
