"""The reference's own two end-to-end tests of this path, re-stated against platipy_amd
(platipy/imaging/tests/test_cardiac.py:74-142 `test_cardiac_service`, :145-237
`test_cardiac_structure_guided_service`): five synthetic sphere cases, four as atlases written to and read back
from NIfTI files, the fifth segmented by run_cardiac_segmentation; acceptance = the reference's Dice thresholds
(> 0.99 unguided, > 0.9 for both structures guided).

`-m gpu`: the reference's data (60 x 128 x 128) and the reference's test settings, unchanged.
CPU suite: the same tests at half size with shorter schedules on the emulated kernels (thresholds relaxed, stated).
"""
import copy

import numpy as np
import pytest
import torch

from tests.helpers import dice

ORIGIN = (320.0, -52.0, 60.0)


def _sphere(shape, radius, centre):
    z, y, x = np.indices(shape)      # the reference calls these x, y, z; the array is [Z][Y][X] either way
    return ((z - centre[0]) / radius) ** 2.0 + ((y - centre[1]) / radius) ** 2.0 + ((x - centre[2]) / radius) ** 2.0 <= 1


def cardiac_data(pa, scale=1):
    """test_cardiac.py:35-71.  scale = 2 halves every length (CPU suite)."""
    shape = (60 // scale, 128 // scale, 128 // scale)
    data = {}
    for i in range(5):
        case_id = str(i + 1).zfill(3)
        big = _sphere(shape, 25 // scale, (30 // scale + i, 64 // scale + i, 64 // scale))
        small = _sphere(shape, 5 // scale + (scale > 1), (30 // scale + i, 60 // scale + i, 60 // scale))
        ct = np.where(big, 1.0, -1000.0)                                   # float64, as the reference's fixture
        sp = ((0.9 + i * 0.01) * scale, (0.9 + i * 0.01) * scale, (2.5 + i * 0.01) * scale)
        data[case_id] = {"CT": pa.image_from_array(ct, sp, ORIGIN), "WHOLEHEART": pa.image_from_array(big.astype(np.uint8), sp, ORIGIN),
                         "SUBSTRUCTURE": pa.image_from_array(small.astype(np.uint8), sp, ORIGIN)}
    return data


def _write_atlases(data, working_path, structures):
    from platipy_amd.io import write_image

    for case, d in data.items():
        (working_path / f"Case_{case}" / "Images").mkdir(parents=True)
        (working_path / f"Case_{case}" / "Structures").mkdir(parents=True)
        write_image(d["CT"], working_path / f"Case_{case}" / "Images" / f"Case_{case}_CROP.nii.gz")
        for s in structures:
            write_image(d[s], working_path / f"Case_{case}" / "Structures" / f"Case_{case}_{s}_CROP.nii.gz")


def _reference_test_settings(pa, cases, working_path, structures, full):
    """The settings edits of test_cardiac.py:95-126 / :173-209."""
    s = copy.deepcopy(pa.projects.cardiac.CARDIAC_SETTINGS_DEFAULTS)
    s["atlas_settings"]["atlas_id_list"] = cases[:-1]
    s["atlas_settings"]["atlas_path"] = str(working_path)
    s["atlas_settings"]["atlas_structure_list"] = list(structures)
    s["atlas_settings"]["auto_crop_atlas"] = False
    s["atlas_settings"]["guide_structure_name"] = "WHOLEHEART"
    s["deformable_registration_settings"]["resolution_staging"] = [8, 4, 2]
    s["deformable_registration_settings"]["iteration_staging"] = [5, 5, 5]
    s["deformable_registration_settings"]["smoothing_sigmas"] = [0, 0, 0]
    s["deformable_registration_settings"]["default_value"] = -1000
    if len(structures) > 1:
        s["structure_guided_registration_settings"]["iteration_staging"] = [5, 5, 5]
        s["structure_guided_registration_settings"]["resolution_staging"] = [6, 3, 1.5]
    s["iar_settings"]["reference_structure"] = None
    s["label_fusion_settings"]["optimal_threshold"] = {k: 0.5 for k in structures}
    s["vessel_spline_settings"] = {"vessel_name_list": [], "vessel_radius_mm_dict": {}, "scan_direction_dict": {},
                                   "stop_condition_type_dict": {}, "stop_condition_value_dict": {}}
    s["postprocessing_settings"]["run_postprocessing"] = False
    s["geometric_segmentation_settings"]["run_geometric_algorithms"] = False
    if not full:   # CPU suite: half-size data (voxels twice as large), shorter linear schedule
        s["linear_registration_settings"].update({"shrink_factors": [4, 2], "number_of_iterations": 15})
        s["auto_crop_target_image_settings"]["expansion_mm"] = [12, 12, 20]
        s["deformable_registration_settings"]["resolution_staging"] = [8, 4]
        s["deformable_registration_settings"]["iteration_staging"] = [5, 5]
        s["structure_guided_registration_settings"]["resolution_staging"] = [8, 4]
        s["structure_guided_registration_settings"]["iteration_staging"] = [5, 5]
    return s


def _run_service(pa, tmp_path, full, min_dice):
    data = cardiac_data(pa, 1 if full else 2)
    cases = list(data.keys())
    _write_atlases(data, tmp_path, ["WHOLEHEART"])
    settings = _reference_test_settings(pa, cases, tmp_path, ["WHOLEHEART"], full)
    infer = cases[-1]
    output, prob = pa.projects.cardiac.run_cardiac_segmentation(data[infer]["CT"], settings=settings)
    assert "WHOLEHEART" in output
    auto = output["WHOLEHEART"]
    assert auto.GetSize() == data[infer]["CT"].GetSize() and auto.tensor.dtype == torch.uint8
    d = dice(auto.numpy(), data[infer]["WHOLEHEART"].numpy())
    assert d > min_dice, d
    return d


def _run_guided_service(pa, tmp_path, full, min_dice):
    data = cardiac_data(pa, 1 if full else 2)
    cases = list(data.keys())
    structures = ["WHOLEHEART", "SUBSTRUCTURE"]
    _write_atlases(data, tmp_path, structures)
    settings = _reference_test_settings(pa, cases, tmp_path, structures, full)
    infer = cases[-1]
    output, _ = pa.projects.cardiac.run_cardiac_segmentation(data[infer]["CT"], data[infer]["WHOLEHEART"], settings=settings)
    assert "WHOLEHEART" in output and "SUBSTRUCTURE" in output
    d_wh = dice(output["WHOLEHEART"].numpy(), data[infer]["WHOLEHEART"].numpy())
    d_ss = dice(output["SUBSTRUCTURE"].numpy(), data[infer]["SUBSTRUCTURE"].numpy())
    assert d_wh > min_dice and d_ss > min_dice, (d_wh, d_ss)
    return d_wh, d_ss


@pytest.mark.gpu
def test_cardiac_service(gpu_backend, tmp_path):
    """test_cardiac.py:74-142 as the reference has it: Dice > 0.99."""
    import platipy_amd as pa

    _run_service(pa, tmp_path, True, 0.99)


@pytest.mark.gpu
def test_cardiac_structure_guided_service(gpu_backend, tmp_path):
    """test_cardiac.py:145-237 as the reference has it: Dice > 0.9 for the whole heart and the sub-structure."""
    import platipy_amd as pa

    _run_guided_service(pa, tmp_path, True, 0.9)


def test_cardiac_service_half_size_cpu(monkeypatch, tmp_path):
    import platipy_amd as pa
    from tests.helpers import install_emu_runtime

    install_emu_runtime(lambda obj, name, value: monkeypatch.setattr(obj, name, value, raising=False))
    _run_service(pa, tmp_path, False, 0.95)      # 2x coarser voxels: one boundary voxel is worth twice the Dice


def test_cardiac_structure_guided_service_half_size_cpu(monkeypatch, tmp_path):
    import platipy_amd as pa
    from tests.helpers import install_emu_runtime

    install_emu_runtime(lambda obj, name, value: monkeypatch.setattr(obj, name, value, raising=False))
    _run_guided_service(pa, tmp_path, False, 0.8)   # the 3-voxel-radius sub-structure at half size: Dice is coarse


def test_cardiac_refuses_out_of_scope_stages(monkeypatch):
    import platipy_amd as pa
    from tests.helpers import install_emu_runtime

    install_emu_runtime(lambda obj, name, value: monkeypatch.setattr(obj, name, value, raising=False))
    img = pa.image_from_array(np.zeros((4, 4, 4), np.float32), (1, 1, 1), (0, 0, 0))
    with pytest.raises(NotImplementedError, match="vessel"):
        pa.projects.cardiac.run_cardiac_segmentation(img)                    # defaults ask for vessel splining
    s = copy.deepcopy(pa.projects.cardiac.CARDIAC_SETTINGS_DEFAULTS)
    s["vessel_spline_settings"]["vessel_name_list"] = []
    with pytest.raises(NotImplementedError, match="geometric"):
        pa.projects.cardiac.run_cardiac_segmentation(img, settings=s)


def test_cardiac_options_cropped_output_and_postprocessing(monkeypatch, tmp_path):
    """return_as_cropped (cardiac/run.py:942-960, 1143-1144), crop_atlas_to_structures (:570-592) and the
    post-processing stage (:1113-1141: largest component, ball closing, overlap correction) on the half-size data."""
    import platipy_amd as pa
    from tests.helpers import install_emu_runtime

    install_emu_runtime(lambda obj, name, value: monkeypatch.setattr(obj, name, value, raising=False))
    data = cardiac_data(pa, 2)
    cases = list(data.keys())
    structures = ["WHOLEHEART", "SUBSTRUCTURE"]
    _write_atlases(data, tmp_path, structures)
    s = _reference_test_settings(pa, cases[:3] + [cases[-1]], tmp_path, structures, False)     # two atlases are enough here
    s["atlas_settings"]["crop_atlas_to_structures"] = True
    s["atlas_settings"]["crop_atlas_expansion_mm"] = (10, 10, 15)
    s["return_as_cropped"] = True
    s["postprocessing_settings"].update({"run_postprocessing": True, "structures_for_binaryfillhole": ["WHOLEHEART", "NOT_THERE"],
                                         "structures_for_overlap_correction": ["WHOLEHEART", "SUBSTRUCTURE"]})
    infer = cases[-1]
    output, prob = pa.projects.cardiac.run_cardiac_segmentation(data[infer]["CT"], settings=s)
    crop = output["CROP_IMAGE"]
    assert crop.GetSize() != data[infer]["CT"].GetSize()                        # the target was cropped ...
    assert output["WHOLEHEART"].GetSize() == crop.GetSize() == prob["WHOLEHEART"].GetSize()   # ... and so are the results
    # cropped results sit where the crop sits: compare with the ground truth cut out at the same place
    off = np.round((np.array(crop.GetOrigin()) - np.array(ORIGIN)) / np.array(crop.GetSpacing())).astype(int)
    sz = crop.GetSize()
    gt = data[infer]["WHOLEHEART"].numpy()[off[2]:off[2] + sz[2], off[1]:off[1] + sz[1], off[0]:off[0] + sz[0]]
    assert dice(output["WHOLEHEART"].numpy(), gt) > 0.9
    # overlap correction made the two structures disjoint; the larger (whole heart) kept the shared voxels
    wh, ss = output["WHOLEHEART"].numpy() > 0, output["SUBSTRUCTURE"].numpy() > 0
    assert not (wh & ss).any() and wh.sum() > 0
    # return_proba_as_contours (cardiac/run.py:945-951): every atlas's propagated contour, one bit per atlas
    s["return_proba_as_contours"] = True
    s["postprocessing_settings"]["run_postprocessing"] = False
    _, enc = pa.projects.cardiac.run_cardiac_segmentation(data[infer]["CT"], settings=s)
    contours = pa.label.binary_decode_image(enc["WHOLEHEART"])
    assert len(contours) == len(s["atlas_settings"]["atlas_id_list"])
    again = pa.label.binary_encode_structure_list(contours)
    np.testing.assert_array_equal(again.numpy(), enc["WHOLEHEART"].numpy())
    assert all(dice(c.numpy(), gt) > 0.85 for c in contours)
