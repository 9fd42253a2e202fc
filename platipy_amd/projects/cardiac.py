"""Atlas-based cardiac segmentation: the harness of platipy/imaging/projects/cardiac/run.py:507-1147
(run_cardiac_segmentation), on the same one-process-per-MI355X skeleton as projects/multiatlas.py.

What is built is the registration / label-fusion path of that function -- atlas loading, target cropping
(from a guide structure, or from the quick similarity registrations), linear registration, the optional
structure-guided demons on distance-map images of the guide structure, intensity demons, iterative atlas
removal, weighted label fusion, paste-back and the connected-component / closing / overlap post-processing.

Out of scope (not on the registration / fusion hot path) and refused loudly rather than
silently skipped: vessel splining (cardiac/run.py:896-905, label/utils + vessel.py), the geometric valve and
conduction-node definitions (:1046-1111), and the nnU-Net front end of run_hybrid_segmentation (:430-504).
Settings keep the reference's schema; CARDIAC_SETTINGS_DEFAULTS carries the reference's values.
"""
import copy
import logging
import os

from ..image import as_image
from .multiatlas import atlas_pipeline

logger = logging.getLogger(__name__)

ATLAS_PATH = os.environ.get("ATLAS_PATH", os.path.join(os.path.expanduser("~"), ".platipy", "cardiac", "test_atlas"))

_VESSELS = ["LANTDESCARTERY", "LCIRCUMFLEXARTERY", "LCORONARYARTERY", "RCORONARYARTERY"]

CARDIAC_SETTINGS_DEFAULTS = {   # cardiac/run.py:75-270
    "atlas_settings": {
        "atlas_id_list": ["03", "05", "08", "10", "11", "12", "13", "16", "24", "35"],
        "atlas_structure_list": [
            "AORTICVALVE", "ASCENDINGAORTA", "LANTDESCARTERY", "LCIRCUMFLEXARTERY", "LCORONARYARTERY", "LEFTATRIUM",
            "LEFTVENTRICLE", "MITRALVALVE", "PULMONARYARTERY", "PULMONICVALVE", "RCORONARYARTERY", "RIGHTATRIUM",
            "RIGHTVENTRICLE", "SVC", "TRICUSPIDVALVE", "WHOLEHEART",
        ],
        "atlas_path": ATLAS_PATH,
        "atlas_image_format": "Case_{0}/Images/Case_{0}_CROP.nii.gz",
        "atlas_label_format": "Case_{0}/Structures/Case_{0}_{1}_CROP.nii.gz",
        "crop_atlas_to_structures": False,
        "crop_atlas_expansion_mm": (20, 20, 40),
        "guide_structure_name": "WHOLEHEART",
        "superior_extension": 30,
    },
    "auto_crop_target_image_settings": {"expansion_mm": [20, 20, 40]},
    "linear_registration_settings": {
        "reg_method": "affine",
        "shrink_factors": [16, 8, 4],
        "smooth_sigmas": [0, 0, 0],
        "sampling_rate": 0.75,
        "default_value": -1000,
        "number_of_iterations": 50,
        "metric": "mean_squares",
        "optimiser": "gradient_descent_line_search",
        "verbose": False,
    },
    "structure_guided_registration_settings": {
        "isotropic_resample": True,
        "resolution_staging": [16, 8, 2],
        "iteration_staging": [50, 50, 50],
        "smoothing_sigmas": [0, 0, 0],
        "ncores": 8,
        "default_value": 0,
        "verbose": False,
    },
    "deformable_registration_settings": {
        "isotropic_resample": True,
        "resolution_staging": [6, 3, 1.5],
        "iteration_staging": [200, 150, 100],
        "smoothing_sigmas": [0, 0, 0],
        "ncores": 8,
        "default_value": 0,
        "verbose": False,
    },
    "iar_settings": {
        "reference_structure": False,
        "smooth_distance_maps": True,
        "smooth_sigma": 1,
        "z_score_statistic": "mad",
        "outlier_method": "iqr",
        "outlier_factor": 1.5,
        "min_best_atlases": 5,
        "project_on_sphere": False,
    },
    "label_fusion_settings": {
        "vote_type": "unweighted",
        "vote_params": None,
        "optimal_threshold": {
            "AORTICVALVE": 0.5, "ASCENDINGAORTA": 0.44, "LEFTATRIUM": 0.40, "LEFTVENTRICLE": 0.45, "MITRALVALVE": 0.5,
            "PULMONARYARTERY": 0.46, "PULMONICVALVE": 0.5, "RIGHTATRIUM": 0.38, "RIGHTVENTRICLE": 0.42, "SVC": 0.44,
            "TRICUSPIDVALVE": 0.5, "WHOLEHEART": 0.5,
        },
    },
    "vessel_spline_settings": {
        "vessel_name_list": list(_VESSELS),
        "vessel_radius_mm_dict": {v: 2 for v in _VESSELS},
        "scan_direction_dict": {v: ("x" if v == "LCORONARYARTERY" else "z") for v in _VESSELS},
        "stop_condition_type_dict": {v: "count" for v in _VESSELS},
        "stop_condition_value_dict": {v: 2 for v in _VESSELS},
    },
    "geometric_segmentation_settings": {
        "run_geometric_algorithms": True,
        "geometric_name_suffix": "_GEOMETRIC",
        "atlas_structure_names": {
            "atlas_left_ventricle": "LEFTVENTRICLE", "atlas_right_ventricle": "RIGHTVENTRICLE", "atlas_left_atrium": "LEFTATRIUM",
            "atlas_right_atrium": "RIGHTATRIUM", "atlas_ascending_aorta": "ASCENDINGAORTA",
            "atlas_pulmonary_artery": "PULMONARYARTERY", "atlas_superior_vena_cava": "SVC", "atlas_whole_heart": "WHOLEHEART",
        },
        "valve_definitions": {
            "mitral_valve_thickness_mm": 10, "mitral_valve_radius_mm": 15, "tricuspid_valve_thickness_mm": 10,
            "tricuspid_valve_radius_mm": 15, "pulmonic_valve_thickness_mm": 10, "aortic_valve_thickness_mm": 10,
        },
        "conduction_system_definitions": {"sinoatrial_node_radius_mm": 10, "atrioventricular_node_radius_mm": 10},
    },
    "postprocessing_settings": {
        "run_postprocessing": True,
        "binaryfillhole_mm": 3,
        "structures_for_binaryfillhole": [
            "ASCENDINGAORTA", "LEFTATRIUM", "LEFTVENTRICLE", "RIGHTATRIUM", "RIGHTVENTRICLE", "SVC", "AORTICVALVE",
            "MITRALVALVE", "PULMONICVALVE", "TRICUSPIDVALVE", "WHOLEHEART",
        ],
        "structures_for_overlap_correction": [
            "ASCENDINGAORTA", "LEFTATRIUM", "LEFTVENTRICLE", "RIGHTATRIUM", "RIGHTVENTRICLE", "PULMONARYARTERY", "SVC",
        ],
    },
    "return_atlas_guide_structure": False,
    "return_as_cropped": False,
    "return_proba_as_contours": False,
}


def run_cardiac_segmentation(img, guide_structure=None, settings=CARDIAC_SETTINGS_DEFAULTS, atlases=None, streams_per_gpu=1,
                             return_atlas_set=False):
    """Runs the atlas-based cardiac segmentation (reference cardiac/run.py:507-1147).

    img: target Image; guide_structure: optional binary Image on img's grid (e.g. a whole-heart mask) that
    switches on target cropping from the structure and structure-guided registration; settings: the reference's
    nested dict.  `atlases` / `streams_per_gpu` as in multiatlas.run_segmentation (atlases are read from
    atlas_settings["atlas_path"] when not given).  Returns (results, results_prob), with return_atlas_set also the per-atlas
    propagated images / labels / weight maps (what the reference keeps in its atlas_set dictionary).
    """
    settings = copy.deepcopy(settings)
    vessels = settings.get("vessel_spline_settings", {}).get("vessel_name_list", [])
    if len(vessels) > 0:
        raise NotImplementedError(
            "run_cardiac_segmentation: vessel splining (vessel_spline_settings['vessel_name_list']) is outside this build's "
            "scope; pass an empty list")
    if settings.get("geometric_segmentation_settings", {}).get("run_geometric_algorithms", False):
        raise NotImplementedError(
            "run_cardiac_segmentation: geometric valve / conduction-node definitions are outside this build's scope; set "
            "geometric_segmentation_settings['run_geometric_algorithms'] = False")
    out = atlas_pipeline(as_image(img), settings, guide_structure, atlases, streams_per_gpu, cardiac=True)
    run_cardiac_segmentation.last_iar_removed = out["iar_removed"]
    if return_atlas_set:
        return out["results"], out["results_prob"], out["atlas_set"]
    return out["results"], out["results_prob"]
