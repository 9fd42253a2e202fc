"""Drop-in for platipy/imaging/label/iar.py:59-301 (run_iar) and label/projection.py:67-92
(evaluate_distance_to_reference): iterative atlas removal.

GPU: consensus fusion (combine_labels), thresholding + largest component (process_probability_image), the
exact Euclidean distance map of every test contour and the reference's label contour.  Host: the statistics
on the sampled distances (<= 1e5 numbers per atlas) -- MAD z-scores, histogram, Gaussian fit, IQR fence --
exactly as the reference computes them with numpy/scipy.  The spherical-projection branch
(project_on_sphere=True; off by default, cardiac/run.py:163) is not implemented.
"""
import logging
import sys

import numpy as np
import torch
from scipy.optimize import curve_fit
from scipy.stats import norm as scipy_norm

from .. import runtime
from ..image import as_image
from .fusion import combine_labels, process_probability_image

logger = logging.getLogger(__name__)


def median_absolute_deviation(data, axis=None):
    return np.median(np.abs(data - np.median(data, axis=axis)), axis=axis)


def gaussian_curve(x, a, m, s):
    return a * scipy_norm.pdf(x, loc=m, scale=s)


def distance_map(mask, signed=True, inside_positive=False):
    """sitk.SignedMaurerDistanceMap(mask, squaredDistance=False, useImageSpacing=True) (sitk.Abs of it when
    signed=False): fp32 Image of distances (mm) to the object's border voxels."""
    mask = as_image(mask)
    ctx = runtime.context(mask.device)
    m = (mask.tensor != 0).to(torch.uint8).contiguous()
    out = torch.empty(mask.shape, dtype=torch.float32, device=m.device)
    ctx.distance_map(m, mask.geom(), out, signed=signed, inside_positive=inside_positive)
    return mask.like(out)


def label_contour(mask):
    """sitk.LabelContour(mask): uint8 Image of the object's face-connected boundary voxels."""
    mask = as_image(mask)
    ctx = runtime.context(mask.device)
    m = (mask.tensor != 0).to(torch.uint8).contiguous()
    out = torch.empty_like(m)
    ctx.label_contour(m, mask.GetSize(), out)
    return mask.like(out)


def evaluate_distance_to_reference(reference_volume, test_volume, resample_factor=1):
    """Distance from every point of the reference surface to the test volume's surface (projection.py:67-92)."""
    test_distance_map = distance_map(test_volume, signed=False)
    ref_surface = label_contour(reference_volume).tensor == 1
    surface_values = test_distance_map.tensor[ref_surface]      # raster order, like numpy boolean indexing
    return surface_values[::resample_factor].cpu().numpy()


def run_iar(atlas_set, reference_structure, smooth_distance_maps=False, smooth_sigma=1, z_score_statistic="MAD",
            outlier_method="IQR", min_best_atlases=10, outlier_factor=1.5, iteration=0, single_step=False,
            project_on_sphere=False, label="DIR"):
    """Perform iterative atlas removal on the atlas_set (reference iar.py:59-301)."""
    if project_on_sphere:
        raise NotImplementedError("project_on_sphere=True is outside this build's scope (SURVEY 2)")
    remaining_id_list = list(atlas_set.keys())
    probability_label = combine_labels(atlas_set, reference_structure, label=label)[reference_structure]

    if len(remaining_id_list) < 12:       # iar.py:104-112 (the second branch is unreachable there too)
        resample_factor = 5
    elif len(remaining_id_list) < 7:
        resample_factor = 10
    else:
        resample_factor = 1

    reference_volume = process_probability_image(probability_label, threshold=0.95)
    g_val_list = []
    for test_id in remaining_id_list:
        test_volume = process_probability_image(atlas_set[test_id][label][reference_structure], 0.1)
        g_val_list.append(evaluate_distance_to_reference(reference_volume, test_volume, resample_factor=resample_factor))

    q_results = {}
    for i, (test_id, g_vals) in enumerate(zip(remaining_id_list, g_val_list)):
        g_val_list_test = g_val_list[:]
        g_val_list_test.pop(i)
        if z_score_statistic.lower() == "std":
            g_val_mean = np.mean(g_val_list_test, axis=0)
            g_val_std = np.std(g_val_list_test, axis=0)
            if np.any(g_val_std == 0):
                g_val_std[g_val_std == 0] = g_val_std.mean()
            z_score_vals_array = (g_vals - g_val_mean) / g_val_std
        elif z_score_statistic.lower() == "mad":
            g_val_median = np.median(g_val_list_test, axis=0)
            g_val_mad = 1.4826 * median_absolute_deviation(g_val_list_test, axis=0)
            if np.any(g_val_mad == 0):
                g_val_mad[g_val_mad == 0] = np.median(g_val_mad)
            z_score_vals_array = (g_vals - g_val_median) / g_val_mad
        else:
            raise ValueError("z_score must be one of: MAD, STD")
        z_score_vals = np.ravel(z_score_vals_array)

        bins = np.linspace(-15, 15, 501)
        z_density, bin_edges = np.histogram(z_score_vals, bins=bins, density=True)
        bin_centers = (bin_edges[1:] + bin_edges[:-1]) / 2.0
        try:
            popt, _ = curve_fit(f=gaussian_curve, xdata=bin_centers, ydata=z_density)
            z_ideal = gaussian_curve(bin_centers, *popt)
            z_diff = np.abs(z_density - z_ideal)
        except (RuntimeError, ValueError):
            z_ideal = gaussian_curve(bin_centers, a=1, m=z_density.mean(), s=z_density.std())
            z_diff = np.abs(z_density - z_ideal)
        trap = np.trapezoid if hasattr(np, "trapezoid") else np.trapz
        q_value = trap(z_diff * np.abs(bin_centers) ** 2, bin_centers)
        q_results[test_id] = np.float64(q_value)

    result_list = [r for r in q_results.values() if ~np.isnan(r) and np.isfinite(r)]
    best_results = np.sort(result_list)[: max([min_best_atlases, len(result_list) - 3])]
    if outlier_method.lower() == "iqr":
        outlier_limit = np.percentile(best_results, 75, axis=0) + outlier_factor * np.subtract(
            *np.percentile(best_results, [75, 25], axis=0))
    elif outlier_method.lower() == "std":
        outlier_limit = np.mean(best_results, axis=0) + outlier_factor * np.std(best_results, axis=0)
    else:
        logger.error(" outlier_method must be one of: IQR, STD")
        sys.exit()

    keep_id_list = [idx for idx, result in q_results.items() if result <= outlier_limit]
    run_iar.last_q_results = dict(q_results)       # diagnostic hook for tests / logging
    if len(keep_id_list) < len(remaining_id_list):
        iteration += 1
        atlas_set_new = {i: atlas_set[i] for i in keep_id_list}
        if single_step:
            return atlas_set_new
        return run_iar(atlas_set=atlas_set_new, reference_structure=reference_structure,
                       smooth_distance_maps=smooth_distance_maps, smooth_sigma=smooth_sigma, z_score_statistic=z_score_statistic,
                       outlier_method=outlier_method, min_best_atlases=min_best_atlases, outlier_factor=outlier_factor,
                       iteration=iteration, project_on_sphere=project_on_sphere, label=label)
    return atlas_set
