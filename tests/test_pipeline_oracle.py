"""The WHOLE multi-atlas pipeline against an independent whole (round 6): platipy_amd.projects.multiatlas.run_segmentation against
oracle/pipeline_oracle.py::run_segmentation -- the reference's run_segmentation (multiatlas/run.py:106-441) restated on the CPU
oracle's restatements of every function it calls, the linear registrations included (oracle/linear_oracle.py::registration).
Until now a11 was held to Dice thresholds only.  Three CT-like atlases (warps of one template, labels carried by the same warps),
the pipeline's own kinds of settings on a small grid: quick similarity registration for the crop, similarity + line search,
isotropic demons levels with sigma 0, local-vote fusion.

Asserted: the same crop box; each atlas's linear map sends the target's corners within 0.1 mm of the oracle's; the fused masks
differ in at most 1 % of the union's voxels (contour voxels: the two chains' registrations agree to a few hundredths of a
millimetre, not bit for bit); probabilities within 0.05 at 99 % of the voxels; Dice with the template's label as good as the
oracle's.  Measured values go to profiles/round6_parity_pipeline_whole.json (tests.helpers.record_stats).

The oracle is parity-unpinned."""
import copy

import numpy as np

from oracle import oracle as O
from oracle import pipeline_oracle as PO
from tests.helpers import dice, phantom, random_dvf, record_stats

INNER, PAD = (32, 56, 64), ((5, 5), (9, 9), (11, 11))
SHAPE, SPACING, ORIGIN = (42, 74, 86), (1.3, 1.3, 2.5), (-40.0, -35.0, 10.0)


def _case():
    template = np.pad(phantom(INNER, seed=900, noise=0), PAD, constant_values=-1000.0)      # a body with air around it
    zz, yy, xx = np.meshgrid(*[np.arange(v) for v in SHAPE], indexing="ij")
    label = (((xx - 44) / 13.0) ** 2 + ((yy - 36) / 11.0) ** 2 + ((zz - 21) / 7.0) ** 2 < 1).astype(np.uint8)
    small = (((xx - 41) / 5.0) ** 2 + ((yy - 34) / 4.0) ** 2 + ((zz - 20) / 3.0) ** 2 < 1).astype(np.uint8)
    rng = np.random.default_rng(901)
    # (noise inside the body only: the air stays at exactly -1000 HU, so that the crop's "mean of the registered atlases >
    # -1000" is the body and the crop box is a real sub-volume -- crop, per-atlas resampling onto it and paste-back all run)
    def noisy(vol):
        return np.where(vol > -990.0, vol + rng.normal(0, 5, SHAPE), -1000.0).astype(np.float32)

    target = noisy(template)
    atlases = {}
    for k in range(3):
        dv = random_dvf(SHAPE, SPACING, seed=910 + 10 * k, max_mm=3.5).astype(np.float64)
        dv[0] += 1.5 * (k - 1)                                     # a residual shift for the linear stage to find
        tv = O.Vol(template, SPACING, ORIGIN)
        ct = O.warp_image(tv, dv, edge_value=-1000.0).arr
        ct = noisy(ct)
        fv = O.Vol(dv, SPACING, ORIGIN)
        lab = O.apply_transform(O.Vol(label, SPACING, ORIGIN), field_vol=fv, default_value=0, interpolator=O.INTERP_NEAREST).arr
        sub = O.apply_transform(O.Vol(small, SPACING, ORIGIN), field_vol=fv, default_value=0, interpolator=O.INTERP_NEAREST).arr
        atlases[f"{k:02d}"] = (ct, lab, sub)
    return target, label, small, atlases


def _settings(ids):
    from platipy_amd.projects.multiatlas import MUTLIATLAS_SETTINGS_DEFAULTS

    s = copy.deepcopy(MUTLIATLAS_SETTINGS_DEFAULTS)
    s["atlas_settings"]["atlas_id_list"] = ids
    s["atlas_settings"]["atlas_structure_list"] = ["HEART", "NODE"]
    s["auto_crop_target_image_settings"]["expansion_mm"] = [1.5, 1.5, 3]       # one voxel: the box stays inside the image
    s["linear_registration_settings"].update({"reg_method": "similarity", "shrink_factors": [4, 2], "smooth_sigmas": [0, 0],
                                              "number_of_iterations": 10})
    s["deformable_registration_settings"].update({"isotropic_resample": True, "resolution_staging": [6, 3, 1.5],
                                                  "iteration_staging": [12, 8, 6], "smoothing_sigmas": [0, 0, 0]})
    s["label_fusion_settings"]["vote_type"] = "local"
    return s


def test_run_segmentation_against_the_oracle_s_whole_pipeline(host_api):
    pa = host_api
    target, label, small, atl = _case()
    ids = sorted(atl)
    st = _settings(ids)
    p_atlases = {i: {"CT Image": pa.image_from_array(ct, SPACING, ORIGIN), "HEART": pa.image_from_array(lab, SPACING, ORIGIN),
                     "NODE": pa.image_from_array(sub, SPACING, ORIGIN)} for i, (ct, lab, sub) in atl.items()}
    o_atlases = {i: {"CT Image": O.Vol(ct, SPACING, ORIGIN), "HEART": O.Vol(lab, SPACING, ORIGIN), "NODE": O.Vol(sub, SPACING, ORIGIN)}
                 for i, (ct, lab, sub) in atl.items()}
    res, prob, aset = pa.projects.multiatlas.run_segmentation(pa.image_from_array(target, SPACING, ORIGIN), st, atlases=p_atlases,
                                                             return_atlas_set=True)
    wres, wprob, rec = PO.run_segmentation(O.Vol(target, SPACING, ORIGIN), st, o_atlases)
    stats = {"crop_box_oracle": [rec["crop_box_size"], rec["crop_box_index"]],
             "crop_box_product": [list(v) for v in pa.projects.multiatlas.run_segmentation.last_crop_box], "structures": {}}
    # the linear stage, atlas by atlas: where the composite sends the corners of the (full) target
    n = np.array(SHAPE[::-1], dtype=np.float64) - 1
    corners = np.array([[i, j, k] for i in (0, n[0]) for j in (0, n[1]) for k in (0, n[2])]) * np.array(SPACING) + np.array(ORIGIN)
    worst = 0.0
    for i in ids:
        tfm = aset[i]["DIR"].get("Linear Transform") if isinstance(aset[i].get("DIR"), dict) else None
        if tfm is None:
            continue
        A, off = tfm.matrix_offset()
        Aw, ow = rec["linear"][i]
        worst = max(worst, float(np.sqrt((((corners @ np.asarray(A).T + off) - (corners @ Aw.T + ow)) ** 2).sum(1)).max()))
    stats["linear_corner_mm_worst"] = worst
    for s, truth in (("HEART", label), ("NODE", small)):
        got, want = res[s].numpy(), wres[s].arr
        union = int(((got > 0) | (want > 0)).sum())
        dp = np.abs(prob[s].numpy().astype(np.float64) - wprob[s].arr)
        stats["structures"][s] = {"voxels_product": int(got.sum()), "voxels_oracle": int(want.sum()), "voxels_differing": int((got != want).sum()),
                                  "union": union, "dice_product_vs_oracle": float(dice(got, want)), "dice_product_vs_template": float(dice(got, truth)),
                                  "dice_oracle_vs_template": float(dice(want, truth)), "prob_abs_diff_p99": float(np.quantile(dp, 0.99)),
                                  "prob_abs_diff_max": float(dp.max())}
    record_stats("pipeline_whole_run_segmentation", stats)
    print("whole pipeline, product vs oracle:", stats)
    assert stats["crop_box_product"] == stats["crop_box_oracle"], stats
    assert int(np.prod(stats["crop_box_oracle"][0])) < int(np.prod(SHAPE)), stats        # a real crop
    assert 0.0 < stats["linear_corner_mm_worst"] <= 0.1, stats
    for s, v in stats["structures"].items():
        assert v["voxels_differing"] <= 0.01 * v["union"] + 2, (s, v)
        assert v["prob_abs_diff_p99"] <= 0.05, (s, v)
        assert v["dice_product_vs_template"] >= v["dice_oracle_vs_template"] - 0.01, (s, v)
    assert stats["structures"]["HEART"]["dice_product_vs_template"] > 0.9
    assert res["HEART"].GetSize() == tuple(SHAPE[::-1])


def test_run_cardiac_segmentation_guided_against_the_oracle_s_whole_pipeline(host_api):
    """The structure-guided cardiac pipeline (cardiac/run.py:507-1147 with a guide structure: crop from the structure,
    registration images from the guide structures' distance maps, linear + structure-guided demons on them, images masked to
    the extended structures, intensity demons, fusion) against oracle/pipeline_oracle.py::run_cardiac_guided, with the switches
    the reference's own guided test sets (no vessels, no geometric definitions, no atlas removal, no post-processing)."""
    from platipy_amd.projects.cardiac import CARDIAC_SETTINGS_DEFAULTS

    pa = host_api
    target, label, small, atl = _case()
    ids = sorted(atl)
    st = copy.deepcopy(CARDIAC_SETTINGS_DEFAULTS)
    st["atlas_settings"].update({"atlas_id_list": ids, "atlas_structure_list": ["HEART", "NODE"], "guide_structure_name": "HEART",
                                 "superior_extension": 10})
    st["auto_crop_target_image_settings"]["expansion_mm"] = [8, 8, 10]
    st["linear_registration_settings"].update({"reg_method": "similarity", "shrink_factors": [4, 2], "smooth_sigmas": [0, 0],
                                               "number_of_iterations": 8})
    st["structure_guided_registration_settings"].update({"resolution_staging": [8, 4, 2], "iteration_staging": [6, 6, 6]})
    st["deformable_registration_settings"].update({"resolution_staging": [6, 3, 1.5], "iteration_staging": [8, 6, 6]})
    st["iar_settings"]["reference_structure"] = None
    st["label_fusion_settings"].update({"vote_type": "local", "optimal_threshold": {"HEART": 0.5, "NODE": 0.5}})
    st["vessel_spline_settings"] = {"vessel_name_list": [], "vessel_radius_mm_dict": {}, "scan_direction_dict": {},
                                    "stop_condition_type_dict": {}, "stop_condition_value_dict": {}}
    st["postprocessing_settings"]["run_postprocessing"] = False
    st["geometric_segmentation_settings"]["run_geometric_algorithms"] = False
    p_atlases = {i: {"CT Image": pa.image_from_array(ct, SPACING, ORIGIN), "HEART": pa.image_from_array(lab, SPACING, ORIGIN),
                     "NODE": pa.image_from_array(sub, SPACING, ORIGIN)} for i, (ct, lab, sub) in atl.items()}
    o_atlases = {i: {"CT Image": O.Vol(ct, SPACING, ORIGIN), "HEART": O.Vol(lab, SPACING, ORIGIN), "NODE": O.Vol(sub, SPACING, ORIGIN)}
                 for i, (ct, lab, sub) in atl.items()}
    res, prob = pa.projects.cardiac.run_cardiac_segmentation(pa.image_from_array(target, SPACING, ORIGIN), pa.image_from_array(label, SPACING, ORIGIN),
                                                            settings=st, atlases=p_atlases)
    wres, wprob, rec = PO.run_cardiac_guided(O.Vol(target, SPACING, ORIGIN), O.Vol(label, SPACING, ORIGIN), st, o_atlases)
    stats = {"crop_box_oracle": [rec["crop_box_size"], rec["crop_box_index"]], "structures": {}}
    assert set(res) == set(wres) == {"HEART", "NODE"}
    np.testing.assert_array_equal(res["HEART"].numpy(), label)          # the guide structure is handed back (cardiac/run.py:994-1004)
    np.testing.assert_array_equal(wres["HEART"].arr, label)
    got, want = res["NODE"].numpy(), wres["NODE"].arr
    dp = np.abs(prob["NODE"].numpy().astype(np.float64) - wprob["NODE"].arr)
    stats["structures"]["NODE"] = {"voxels_product": int(got.sum()), "voxels_oracle": int(want.sum()), "voxels_differing": int((got != want).sum()),
                                   "dice_product_vs_oracle": float(dice(got, want)), "dice_product_vs_template": float(dice(got, small)),
                                   "dice_oracle_vs_template": float(dice(want, small)), "prob_abs_diff_p99": float(np.quantile(dp, 0.99)),
                                   "prob_abs_diff_max": float(dp.max())}
    record_stats("pipeline_whole_run_cardiac_guided", stats)
    print("guided cardiac pipeline, product vs oracle:", stats)
    v = stats["structures"]["NODE"]
    assert v["voxels_differing"] <= 0.02 * max(v["voxels_product"], v["voxels_oracle"]) + 2, v
    assert v["prob_abs_diff_p99"] <= 0.05 and v["dice_product_vs_template"] >= v["dice_oracle_vs_template"] - 0.02 and v["dice_product_vs_template"] > 0.8, v
