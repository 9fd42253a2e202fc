from .fusion import combine_labels, compute_weight_map, process_probability_image  # noqa: F401
