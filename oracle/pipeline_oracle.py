"""oracle/pipeline_oracle.py -- TEST INFRASTRUCTURE ONLY (see oracle/pp_oracle.h; PARITY UNPINNED).

platipy/imaging/projects/multiatlas/run.py:106-441 (run_segmentation, the unguided multi-atlas pipeline) restated line by line on the
CPU oracle's own restatements of the functions it calls: linear_registration (oracle/linear_oracle.py::registration + the resample
at registration/linear.py:241-258), apply_transform, fast_symmetric_forces_demons_registration, compute_weight_map, combine_labels,
process_probability_image (oracle/oracle.py), label_to_roi / crop_to_roi (utils/crop.py:24-77) and sitk.Paste.  Atlases are
passed in memory ({atlas_id: {"CT Image": Vol, "<structure>": Vol}}) instead of being read from NIfTI files (run.py:155-170).
Used by tests/test_pipeline_oracle.py to hold the PRODUCT's whole run_segmentation to an independent whole."""
import numpy as np

from oracle import linear_oracle as LO
from oracle import oracle as O

LINEAR, NEAREST = O.INTERP_LINEAR, O.INTERP_NEAREST


def label_to_roi(label, expansion_mm=(0, 0, 0)):
    """utils/crop.py:24-70: bounding box (x, y, z) of label > 0 (the sum of several), expanded and clipped."""
    labels = label if isinstance(label, (list, tuple)) else [label]
    ref = labels[0]
    mask = sum(l.arr.astype(np.float64) for l in labels) > 0
    zz, yy, xx = np.nonzero(mask)
    index = np.array([xx.min(), yy.min(), zz.min()])
    size = np.array([xx.max() - xx.min() + 1, yy.max() - yy.min() + 1, zz.max() - zz.min() + 1])
    expansion = (np.array(expansion_mm) / np.array(ref.spacing)).astype(int)
    crop_index = np.max([index - expansion, np.array([0, 0, 0])], axis=0)
    crop_size = np.min([np.array(ref.size) - crop_index, size + 2 * expansion], axis=0)
    return [int(v) for v in crop_size], [int(v) for v in crop_index]


def crop_to_roi(vol, size, index):
    """sitk.RegionOfInterest(image, size, index): the sub-volume with its own origin."""
    x0, y0, z0 = index
    nx, ny, nz = size
    D = np.asarray(vol.direction, dtype=np.float64).reshape(3, 3)
    origin = np.asarray(vol.origin) + D @ (np.asarray(index, dtype=np.float64) * np.asarray(vol.spacing))
    return O.Vol(vol.arr[..., z0:z0 + nz, y0:y0 + ny, x0:x0 + nx].copy(), vol.spacing, tuple(origin), vol.direction)


def paste(template, vol, index):
    """sitk.Paste(template, source, source.GetSize(), (0, 0, 0), index)"""
    out = template.arr.copy()
    nz, ny, nx = vol.arr.shape
    out[index[2]:index[2] + nz, index[1]:index[1] + ny, index[0]:index[0] + nx] = vol.arr.astype(out.dtype)
    return template.like(out)


def linear_registration(fixed, moving, reg_method="similarity", metric="mean_squares", optimiser="gradient_descent", shrink_factors=(8, 2, 1),
                        smooth_sigmas=(4, 2, 0), sampling_rate=0.25, final_interp=2, number_of_iterations=50, default_value=None,
                        verbose=False, **_ignored):
    """registration/linear.py:50-260 -> (registered image, (A, off)) with (A, off) the composite [initial, optimised] as q = A p + off."""
    f32 = fixed.like(fixed.arr.astype(np.float32))
    m32 = moving.like(moving.arr.astype(np.float32))
    reg = LO.registration(f32, m32, reg_method, optimiser, list(shrink_factors), list(smooth_sigmas), sampling_rate, number_of_iterations,
                          metric=metric)
    A, off = reg["matrix_offset"]
    if default_value is None:
        default_value = -1000 if float(m32.arr.min()) <= -1000 else 0
    interp = {1: NEAREST, 2: LINEAR}[int(final_interp)]
    registered = O.apply_transform(m32, fixed, affine=(A, off), default_value=default_value, interpolator=interp)
    return fixed.like(O._cast(registered.arr, moving.arr.dtype)), (A, off)


QUICK_REG_SETTINGS = dict(reg_method="similarity", shrink_factors=[8], smooth_sigmas=[0], sampling_rate=0.75, default_value=-1000,
                          number_of_iterations=25, final_interp=2, metric="mean_squares", optimiser="gradient_descent_line_search")


def run_segmentation(img, settings, atlases):
    """-> (results, results_prob, record) -- record: crop box, per-atlas transforms, for the test's diagnostics."""
    a_set = settings["atlas_settings"]
    atlas_id_list, atlas_structure_list = list(a_set["atlas_id_list"]), list(a_set["atlas_structure_list"])
    atlas_set = {}
    for atlas_id in atlas_id_list:                                   # run.py:155-190
        image = atlases[atlas_id]["CT Image"]
        structures = {s: atlases[atlas_id][s] for s in atlas_structure_list}
        if a_set.get("crop_atlas_to_structures", False):
            size, index = label_to_roi(list(structures.values()), expansion_mm=a_set["crop_atlas_expansion_mm"])
            image = crop_to_roi(image, size, index)
            structures = {s: crop_to_roi(v, size, index) for s, v in structures.items()}
        atlas_set[atlas_id] = {"Original": {"CT Image": image, **structures}}

    # step 1: automatic cropping (run.py:203-249)
    expansion_mm = settings["auto_crop_target_image_settings"]["expansion_mm"]
    registered = []
    for atlas_id in atlas_id_list[: min(8, len(atlas_id_list))]:
        reg_image, _ = linear_registration(img, atlas_set[atlas_id]["Original"]["CT Image"], **QUICK_REG_SETTINGS)
        registered.append(reg_image.arr.astype(np.float32))
    acc = registered[0]
    for r in registered[1:]:
        acc = acc + r                                                # sum() of sitk float32 images: a left fold
    combined = img.like(((acc / np.float32(len(registered))) > -1000).astype(np.uint8))
    crop_box_size, crop_box_index = label_to_roi(combined, expansion_mm=expansion_mm)
    img_crop = crop_to_roi(img, crop_box_size, crop_box_index)
    record = {"crop_box_size": crop_box_size, "crop_box_index": crop_box_index, "linear": {}}

    # steps 2-3, per atlas (run.py:255-347)
    lin_set = settings["linear_registration_settings"]
    dir_set = {k: v for k, v in settings["deformable_registration_settings"].items() if k not in ("ncores", "verbose")}
    for atlas_id in atlas_id_list:
        orig = atlas_set[atlas_id]["Original"]
        _, (A, off) = linear_registration(img_crop, orig["CT Image"], **lin_set)
        record["linear"][atlas_id] = (A, off)
        rir = {"CT Image": O.apply_transform(orig["CT Image"], img_crop, affine=(A, off), default_value=-1000, interpolator=LINEAR)}
        for s in atlas_structure_list:
            rir[s] = O.apply_transform(orig[s], img_crop, affine=(A, off), default_value=0, interpolator=NEAREST)
        _, dvf, _ = O.fast_symmetric_forces_demons_registration(img_crop, rir["CT Image"], **dir_set)
        out = {"CT Image": O.apply_transform(rir["CT Image"], field_vol=dvf, default_value=-1000, interpolator=LINEAR)}
        for s in atlas_structure_list:
            out[s] = O.apply_transform(rir[s], field_vol=dvf, default_value=0, interpolator=NEAREST)
        atlas_set[atlas_id] = {"DIR": out}

    # step 4: label fusion (run.py:351-362)
    fus = settings["label_fusion_settings"]
    for atlas_id in atlas_id_list:
        d = atlas_set[atlas_id]["DIR"]
        d["Weight Map"] = O.compute_weight_map(img_crop, d["CT Image"], vote_type=fus["vote_type"], vote_params=fus["vote_params"])
    combined_label_dict = O.combine_labels(atlas_set, atlas_structure_list)

    # step 6: threshold, paste back (run.py:367-404)
    results, results_prob = {}, {}
    template_binary = img.like(np.zeros(img.arr.shape, dtype=np.uint8))
    template_prob = img.like(np.zeros(img.arr.shape, dtype=np.float64))
    for s in atlas_structure_list:
        prob = combined_label_dict[s]
        thr = fus["optimal_threshold"].get(s, 0.5)
        results[s] = paste(template_binary, O.process_probability_image(prob, thr), crop_box_index)
        results_prob[s] = paste(template_prob, prob, crop_box_index)
    return results, results_prob, record


# =================================================================================================================================
# platipy/imaging/projects/cardiac/run.py:507-1147 with a guide structure (the structure-guided path), restated the same way; vessel
# splining, geometric valve / node definitions, iterative atlas removal and post-processing are left to their own tests (the caller's
# settings switch them off, as the reference's own test does: platipy/imaging/tests/test_cardiac.py:174-196).


def convert_mask_to_reg_structure(mask, expansion=2):
    """registration/utils.py:302-344: ball dilation by int(expansion / spacing) voxels, inside distance map, zero outside, / max."""
    grown = O.binary_dilate_ball(mask.like((mask.arr != 0).astype(np.uint8)), [int(expansion / s) for s in mask.spacing])
    dm = O.maurer_distance_map(grown, signed=True, inside_positive=True).arr.astype(np.float64) * (grown.arr != 0)
    return mask.like(dm / dm.max())


def extend_mask(mask, extension_mm=10, interior_mm_shape=10):
    """generation/mask.py:107-159, direction ("ax", "sup"): the superior end of the label is continued upwards by the union of its
    last few slices."""
    arr = (mask.arr != 0).astype(mask.arr.dtype)
    occupied = np.nonzero(arr.any(axis=(1, 2)))[0]
    top = int(occupied.max())
    n_ext, n_est = int(extension_mm / mask.spacing[2]), int(interior_mm_shape / mask.spacing[2])
    stop = min(arr.shape[0], top + 1 + n_ext)
    for z in range(top + 1 - n_est, stop):
        arr[z] = arr[top - n_est:top].max(axis=0)      # (re-read every pass, as the reference's loop does: later slices see earlier writes)
    return mask.like(arr)


def _mask_outside(vol, mask, outside):
    """sitk.Mask(image, mask, outsideValue)"""
    return vol.like(np.where(mask != 0, vol.arr, np.asarray(outside, dtype=vol.arr.dtype)))


def run_cardiac_guided(img, guide_structure, settings, atlases):
    a_set = settings["atlas_settings"]
    atlas_id_list, atlas_structure_list = list(a_set["atlas_id_list"]), list(a_set["atlas_structure_list"])
    name = a_set["guide_structure_name"]
    ext = a_set["superior_extension"]
    # step 1 (cardiac/run.py:603-616)
    crop_box_size, crop_box_index = label_to_roi(guide_structure, expansion_mm=settings["auto_crop_target_image_settings"]["expansion_mm"])
    img_crop = crop_to_roi(img, crop_box_size, crop_box_index)
    guide = crop_to_roi(guide_structure, crop_box_size, crop_box_index)
    target_reg_structure = convert_mask_to_reg_structure(guide, 2)
    expanded_target = extend_mask(guide, ext, ext / 2)
    lin_set = settings["linear_registration_settings"]
    strip = ("ncores", "verbose")
    sg_set = {k: v for k, v in settings["structure_guided_registration_settings"].items() if k not in strip}
    dir_set = {k: v for k, v in settings["deformable_registration_settings"].items() if k not in strip}
    atlas_set, record = {}, {"crop_box_size": crop_box_size, "crop_box_index": crop_box_index}
    for atlas_id in atlas_id_list:
        orig = atlases[atlas_id]
        # step 2 (:668-745): the guide structures' registration images are registered, not the CTs
        atlas_reg = convert_mask_to_reg_structure(orig[name], 2)
        _, (A, off) = linear_registration(target_reg_structure, atlas_reg, **lin_set)
        reg_mask = O.apply_transform(atlas_reg, img_crop, affine=(A, off), default_value=0, interpolator=LINEAR)
        expanded = O.apply_transform(extend_mask(orig[name], ext, ext / 2), img_crop, affine=(A, off), default_value=0, interpolator=NEAREST)
        cur = {"CT Image": O.apply_transform(orig["CT Image"], img_crop, affine=(A, off), default_value=-1000, interpolator=LINEAR)}
        for s in atlas_structure_list:
            cur[s] = O.apply_transform(orig[s], img_crop, affine=(A, off), default_value=0, interpolator=NEAREST)
        # step 3a (:751-799): structure-guided demons on the registration images
        _, sg, _ = O.fast_symmetric_forces_demons_registration(target_reg_structure, reg_mask, **sg_set)
        nxt = {"CT Image": O.apply_transform(cur["CT Image"], field_vol=sg, default_value=-1000, interpolator=LINEAR)}
        expanded = O.apply_transform(expanded, img_crop, field_vol=sg, default_value=0, interpolator=NEAREST)
        for s in atlas_structure_list:
            nxt[s] = O.apply_transform(cur[s], field_vol=sg, default_value=0, interpolator=NEAREST)
        # step 3b (:818-870): both images masked to the union of the extended guide structures and to the ATLAS image's soft tissue
        combined = np.maximum(expanded.arr, expanded_target.arr.astype(expanded.arr.dtype))
        atlas_img = _mask_outside(nxt["CT Image"], combined, -1000)
        atlas_img = _mask_outside(atlas_img, atlas_img.arr > -400, -1000)
        target_img = _mask_outside(img_crop, combined, -1000)
        target_img = _mask_outside(target_img, atlas_img.arr > -400, -1000)
        _, dvf, _ = O.fast_symmetric_forces_demons_registration(target_img, atlas_img, **dir_set)
        out = {"CT Image": O.apply_transform(nxt["CT Image"], field_vol=dvf, default_value=-1000, interpolator=LINEAR)}
        for s in atlas_structure_list:
            out[s] = O.apply_transform(nxt[s], field_vol=dvf, default_value=0, interpolator=NEAREST)
        atlas_set[atlas_id] = {"DIR": out}
    # step 5 (:910-922) and 6 (:928-1004)
    fus = settings["label_fusion_settings"]
    for atlas_id in atlas_id_list:
        d = atlas_set[atlas_id]["DIR"]
        d["Weight Map"] = O.compute_weight_map(img_crop, d["CT Image"], vote_type=fus["vote_type"], vote_params=fus["vote_params"])
    combined_label_dict = O.combine_labels(atlas_set, atlas_structure_list)
    results, results_prob = {}, {}
    template_binary = img.like(np.zeros(img.arr.shape, dtype=np.uint8))
    template_prob = img.like(np.zeros(img.arr.shape, dtype=np.float64))
    for s in [k for k in fus["optimal_threshold"] if k in atlas_structure_list]:
        prob = combined_label_dict[s]
        results[s] = paste(template_binary, O.process_probability_image(prob, fus["optimal_threshold"][s]), crop_box_index)
        results_prob[s] = paste(template_prob, prob, crop_box_index)
        if not settings.get("return_atlas_guide_structure", False):
            results[name] = paste(template_binary, guide, crop_box_index)
    return results, results_prob, record
