from .fusion import combine_labels, compute_weight_map, process_probability_image  # noqa: F401
from .iar import distance_map, evaluate_distance_to_reference, label_contour, run_iar  # noqa: F401
from . import utils  # noqa: F401
from .utils import (  # noqa: F401
    binary_decode_image, binary_dilate, binary_encode_structure_list, binary_erode, binary_morphological_closing,
    correct_volume_overlap, largest_component)
