from .fusion import combine_labels, compute_weight_map, process_probability_image  # noqa: F401
from .iar import distance_map, evaluate_distance_to_reference, label_contour, run_iar  # noqa: F401
from .utils import correct_volume_overlap  # noqa: F401
