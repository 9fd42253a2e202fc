"""platipy_amd -- MI355X-native deformable registration and multi-atlas label fusion.

Drop-in for the hot path of pyplati/platipy: `linear_registration`,
`fast_symmetric_forces_demons_registration`, `apply_transform`, `smooth_and_resample` and the
`label.fusion` utilities keep the reference's signatures; volumes live in HBM as torch tensors and
every voxel-level operation runs in hand-written HIP kernels behind the C ABI of
include/platipy_amd.h.  There is no CPU fallback.
"""
import os as _os

# One hardware queue per worker stream.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of
# them the null stream's): with config 5's four atlas chains on four HIP streams two chains shared a queue and ran one after
# the other -- 240 ms of a 269 ms span on that queue alone (profiles/round5_streams_timeline_before.md).  Read when the
# runtime initialises the device, i.e. at the first CUDA call, which comes after this import; an explicit setting wins.
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"      # (process-wide: every HIP user of this process gets 8 queues per device)
    import sys as _sys

    # (torch.cuda.is_initialized() is a flag of torch's, not a HIP call: nothing here may touch the runtime -- a device-count
    # query such as torch.cuda.is_available() would initialise it with the default of 4 queues before the line above counts)
    _torch = _sys.modules.get("torch")
    if _torch is not None and _torch.cuda.is_initialized():
        import warnings as _warnings

        _warnings.warn("platipy_amd was imported after the HIP runtime had initialised: GPU_MAX_HW_QUEUES=8 cannot take effect any "
                       "more, and run_segmentation(streams_per_gpu >= 4) will share hardware queues between atlas chains (results are "
                       "the same; config 5's per-GPU shape runs slower).  Import platipy_amd first or export the variable.")

from .image import Image, image_from_array, array_from_image  # noqa: E402,F401
from .transform import (  # noqa: E402,F401
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
