"""Committed golden vectors (tests/golden/hotpath_small.npz, made by tests/golden/make_golden.py).  BUILD-DERIVED from
the oracle, not from the reference (see that script's header): they keep the oracle honest across edits and give the
product a second, frozen target."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from platipy_amd import _lib
from tests.golden.make_golden import METRIC_MAP, NOTCHED, ORIGIN, SHAPE, SPACING, inputs

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "hotpath_small.npz"))
SIZE = (SHAPE[2], SHAPE[1], SHAPE[0])
FLT_MAX = float(np.finfo(np.float32).max)


def test_generator_inputs_are_reproducible():
    fixed, moving, field, mask = inputs()
    for k, v in (("fixed", fixed), ("moving", moving), ("field", field), ("mask", mask)):
        np.testing.assert_array_equal(G[k], v)


def test_oracle_reproduces_golden():
    vf, vm = O.Vol(G["fixed"], SPACING, ORIGIN), O.Vol(G["moving"], SPACING, ORIGIN)
    f64 = G["field"].astype(np.float64)
    np.testing.assert_array_equal(O.gaussian_operator(1.0, 0.1, 30), G["taps_var1_err0p1"])
    np.testing.assert_array_equal(O.gaussian_operator(2.25, 0.1, 30), G["taps_var2p25_err0p1"])
    np.testing.assert_array_equal(O.gaussian_operator(4.0, 0.01, 32), G["taps_var4_err0p01"])
    np.testing.assert_array_equal(O.discrete_gaussian(vf, 4.0).arr, G["discrete_gaussian_var4"])
    np.testing.assert_array_equal(O.warp_image(vm, f64).arr, G["warp_sentinel"])
    upd, st = O.esm_update(vf, O.Vol(G["warp_sentinel"], SPACING, ORIGIN))
    np.testing.assert_allclose(upd, G["esm_update"], rtol=0, atol=1e-7)   # stored as fp32
    np.testing.assert_allclose([st.metric, st.rms_change, st.n_pixels], G["esm_stats"], rtol=1e-12)
    np.testing.assert_array_equal(O.resample(O.Vol(G["mask"], SPACING, ORIGIN), O.Vol(G["mask"], SPACING, ORIGIN),
                                             field_vol=O.Vol(f64, SPACING, ORIGIN), interp=O.INTERP_NEAREST).arr, G["mask_nn_through_field"])
    np.testing.assert_array_equal(O.label_contour(O.Vol(G["mask"], SPACING, ORIGIN)).arr, G["label_contour"])
    mv = O.Vol(G["mask"], SPACING, ORIGIN)
    np.testing.assert_array_equal(O.binary_dilate_ball(mv, (2, 2, 1)).arr, G["dilate_ball_221"])
    np.testing.assert_array_equal(O.binary_erode_ball(mv, (1, 1, 1)).arr, G["erode_ball_111"])
    np.testing.assert_array_equal(O.binary_closing_ball(O.Vol(NOTCHED(G["mask"]), SPACING, ORIGIN), (2, 1, 0)).arr, G["close_ball_210"])
    from oracle import linear_oracle

    np.testing.assert_allclose(linear_oracle.meansq_affine(G["fixed"], G["moving"], *METRIC_MAP), G["meansq_affine"], rtol=1e-12)


def test_product_morphology_and_metric_match_golden(backend):
    ctx = backend.ctx
    mask = backend.dev(G["mask"])
    for key, radius, op, src in (("dilate_ball_221", (2, 2, 1), 0, mask), ("erode_ball_111", (1, 1, 1), 1, mask),
                                 ("close_ball_210", (2, 1, 0), 2, backend.dev(NOTCHED(G["mask"])))):
        out = backend.empty(SHAPE, np.uint8)
        ctx.binary_morph_ball(src, SIZE, radius, op, out)
        np.testing.assert_array_equal(backend.host(out), G[key])
    assert (G["close_ball_210"] != NOTCHED(G["mask"])).any()            # the closing did close something
    Af, bf, Am, bm, vsize, stride = METRIC_MAP
    want = G["meansq_affine"]
    got = np.array(ctx.meansq_affine(backend.dev(G["fixed"]), SIZE, backend.dev(G["moving"]), SIZE, Af.ravel(), bf, Am.ravel(), bm, vsize, stride))
    assert got[1] == want[1] and want[1] > 50
    np.testing.assert_allclose(got[0], want[0], rtol=1e-5)
    np.testing.assert_allclose(got[2:], want[2:], rtol=2e-4, atol=1e-3 * np.abs(want[2:]).max())
    vals = ctx.metric_values_affine(0, backend.dev(G["fixed"]), SIZE, backend.dev(G["moving"]), SIZE, Af.ravel(), bf, [Am], [bm], vsize, stride)
    assert vals[0, 1] == want[1]
    np.testing.assert_allclose(vals[0, 0], want[0], rtol=1e-5)


def test_product_matches_golden(backend):
    ctx = backend.ctx
    g = _lib.make_geom(SIZE, SPACING, ORIGIN)
    fixed, moving, field, mask = (backend.dev(G[k]) for k in ("fixed", "moving", "field", "mask"))
    for key, var, err, mkw in (("taps_var1_err0p1", 1.0, 0.1, 30), ("taps_var4_err0p01", 4.0, 0.01, 32)):
        np.testing.assert_array_equal(np.array(_lib.gauss_taps(var, err, mkw, lib=backend.lib)), G[key])
    out = backend.empty(SHAPE)
    ctx.discrete_gaussian(fixed, out, SIZE, SPACING, (4.0, 4.0, 4.0), 0.01, 32, True)
    np.testing.assert_allclose(backend.host(out), G["discrete_gaussian_var4"], rtol=0, atol=2e-3)
    f = backend.dev(G["field"])
    ctx.smooth_field(f, SIZE, [1.5 / s for s in SPACING])
    np.testing.assert_allclose(backend.host(f), G["smooth_field"], rtol=0, atol=2e-6)
    ctx.warp(moving, field, g, FLT_MAX, out)
    w = backend.host(out)
    assert ((w == FLT_MAX) == (G["warp_sentinel"] == FLT_MAX)).all()
    ok = w != FLT_MAX
    np.testing.assert_allclose(w[ok], G["warp_sentinel"][ok], rtol=0, atol=5e-3)
    p = ctx.default_demons_params()
    p.smooth_update, p.iterations, p.max_rms_error = 1, 3, 0.0
    p.sigma_d_vox[:] = [1.5 / s for s in SPACING]
    upd = backend.empty((3,) + SHAPE)
    st = ctx.demons_force(fixed, backend.dev(G["warp_sentinel"]), g, p, upd)
    np.testing.assert_allclose(backend.host(upd), G["esm_update"], rtol=2e-5, atol=2e-6)
    assert st.n_pixels == int(G["esm_stats"][2])
    np.testing.assert_allclose([st.metric, st.rms_change], G["esm_stats"][:2], rtol=1e-5)
    for variant in (_lib.DEMONS_FUSED, _lib.DEMONS_STAGED):
        p.variant = variant
        d = backend.empty((3,) + SHAPE)
        st = ctx.demons_execute(fixed, moving, g, p, d)
        np.testing.assert_allclose(backend.host(d), G["execute_3it"], rtol=0, atol=1e-3)
        assert st.elapsed_iterations == int(G["execute_stats"][2])
        np.testing.assert_allclose([st.metric, st.rms_change], G["execute_stats"][:2], rtol=1e-4)
    f = backend.dev(G["field"])
    ctx.recursive_gaussian_field(f, g, [1.5 / s for s in SPACING])
    np.testing.assert_allclose(backend.host(f), G["recursive_gaussian"], rtol=0, atol=3e-6)
    m = backend.empty(SHAPE, np.uint8)
    ctx.resample(mask, g, g, m, field=field, interp=_lib.INTERP_NEAREST, default_value=0, u8=True)
    np.testing.assert_array_equal(backend.host(m), G["mask_nn_through_field"])          # bit-exact
    ctx.weight_map_local(fixed, moving, SIZE, SPACING, 2.0, 1e-5, out)
    np.testing.assert_allclose(backend.host(out), G["weight_local"], rtol=3e-5, atol=0)
    ctx.distance_map(mask, g, out, signed=True)
    np.testing.assert_allclose(backend.host(out), G["distance_map_signed"], rtol=2e-6, atol=2e-5)
    ctx.label_contour(mask, SIZE, m)
    np.testing.assert_array_equal(backend.host(m), G["label_contour"])


# --------------------------------------------------------------------------------------
# REFERENCE vectors: tests/golden/sitk_*.npz, written by `tools/compare_with_sitk.py --emit` wherever SimpleITK exists
# (tools/sitk_vectors.py).  None is committed yet -- SimpleITK is in neither the build image nor the GPU box -- so the
# reference-pinned tests below SKIP LOUDLY and parity stays "unpinned" (DESIGN section 3).  The day a file lands they hold
# the oracle AND the product to SimpleITK's own numbers with no code change.  The same checks run in the CPU suite on a
# file produced through tests/sitk_double (backed by the oracle): that exercises the emit -> load -> compare path, and
# proves nothing about ITK.

import glob  # noqa: E402

REFERENCE_FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sitk_*.npz")))
NO_REFERENCE = ("NO REFERENCE VECTORS (tests/golden/sitk_*.npz): parity with SimpleITK is UNPINNED.  Run "
                "`python tools/compare_with_sitk.py --emit tests/golden/sitk_<version>.npz` where SimpleITK is installed and commit the file.")

# Stated tolerances.  Oracle (fp64 field, ITK's own intermediate precisions) vs SimpleITK: what separate compilations of the
# same arithmetic may differ by.  Product (fp32) vs SimpleITK: DESIGN section 3's fp32 tolerances.
ORACLE_TOL = {"field": 1e-6, "image": 2e-3, "recursive": 1e-6, "distance": 2e-5, "stats_rel": 1e-6}
PRODUCT_TOL = {"field_max": 2e-3, "field_rms": 5e-5, "image": 5e-3, "recursive": 3e-6, "distance_rel": 2e-6, "distance_abs": 2e-5,
               "stats_rel": 1e-4}


def _sitk_demons_oracle(R, key):
    from tools.sitk_vectors import SIGMA

    elapsed, metric, rms, n, max_rms = R[key + "_stats"]
    flt = O.DemonsFilter()
    flt.SetSmoothUpdateField(True)
    flt.SetSmoothDisplacementField(True)
    flt.SetStandardDeviations(SIGMA)
    flt.SetNumberOfIterations(int(n))
    if not np.isnan(max_rms):
        flt.SetMaximumRMSError(float(max_rms))
    got = flt.Execute(O.Vol(R["fixed"], SPACING, ORIGIN), O.Vol(R["moving"], SPACING, ORIGIN)).arr
    return got, flt


def check_oracle_against_reference(R):
    """The oracle's restatement of every stage against the vectors of one sitk_*.npz."""
    from tools.sitk_vectors import SIGMA

    for k, v in zip(("fixed", "moving", "field", "mask"), inputs()):
        np.testing.assert_array_equal(R[k], v)                       # the file was made from this repo's seeded inputs
    f64 = R["field"].astype(np.float64)
    for key in ("execute_1it", "execute_4it", "execute_halt"):
        got, flt = _sitk_demons_oracle(R, key)
        elapsed, metric, rms = R[key + "_stats"][:3]
        assert flt.GetElapsedIterations() == int(elapsed), key           # same Halt() decision
        np.testing.assert_allclose(got, R[key], rtol=0, atol=ORACLE_TOL["field"], err_msg=key)
        np.testing.assert_allclose([flt.GetMetric(), flt.GetRMSChange()], [metric, rms], rtol=ORACLE_TOL["stats_rel"], err_msg=key)
    np.testing.assert_allclose(O.recursive_gaussian_vec(O.Vol(f64, SPACING, ORIGIN), SIGMA).arr, R["recursive_gaussian"], rtol=0,
                               atol=ORACLE_TOL["recursive"])
    vf, vm, vk = O.Vol(R["fixed"], SPACING, ORIGIN), O.Vol(R["moving"], SPACING, ORIGIN), O.Vol(R["mask"], SPACING, ORIGIN)
    np.testing.assert_allclose(O.discrete_gaussian(vf, 4.0).arr, R["discrete_gaussian_var4"], rtol=0, atol=ORACLE_TOL["image"])
    np.testing.assert_allclose(O.discrete_gaussian(vf, 1.0).arr, R["discrete_gaussian_var1"], rtol=0, atol=ORACLE_TOL["image"])
    fv = O.Vol(f64, SPACING, ORIGIN)
    np.testing.assert_allclose(O.resample(vm, vm, field_vol=fv, interp=O.INTERP_LINEAR, default_value=-1000.0).arr, R["resample_linear"],
                               rtol=0, atol=ORACLE_TOL["image"])
    np.testing.assert_array_equal(O.resample(vk, vk, field_vol=fv, interp=O.INTERP_NEAREST, default_value=0).arr, R["resample_nearest"])
    np.testing.assert_allclose(O.maurer_distance_map(vk, signed=True).arr, R["distance_map_signed"], rtol=0, atol=ORACLE_TOL["distance"])
    np.testing.assert_array_equal(O.label_contour(vk).arr, R["label_contour"])


def check_product_against_reference(R, backend):
    """The HIP kernels behind the C ABI against the same vectors (fp32 tolerances of DESIGN section 3)."""
    from tools.sitk_vectors import SIGMA

    ctx = backend.ctx
    g = _lib.make_geom(SIZE, SPACING, ORIGIN)
    fixed, moving, field, mask = (backend.dev(R[k]) for k in ("fixed", "moving", "field", "mask"))
    for key in ("execute_1it", "execute_4it", "execute_halt"):
        elapsed, metric, rms, n, max_rms = R[key + "_stats"]
        for variant in (_lib.DEMONS_FUSED, _lib.DEMONS_STAGED):
            p = ctx.default_demons_params()
            p.smooth_update, p.smooth_displacement, p.iterations, p.variant = 1, 1, int(n), variant
            p.sigma_d_vox[:] = SIGMA
            if not np.isnan(max_rms):
                p.max_rms_error = float(max_rms)
            d = backend.empty((3,) + SHAPE)
            st = ctx.demons_execute(fixed, moving, g, p, d)
            err = np.abs(backend.host(d) - R[key])
            assert st.elapsed_iterations == int(elapsed), (key, variant)
            assert err.max() <= PRODUCT_TOL["field_max"] and np.sqrt((err ** 2).mean()) <= PRODUCT_TOL["field_rms"], (key, variant, err.max())
            np.testing.assert_allclose([st.metric, st.rms_change], [metric, rms], rtol=PRODUCT_TOL["stats_rel"])
    f = backend.dev(R["field"])
    ctx.recursive_gaussian_field(f, g, SIGMA)
    np.testing.assert_allclose(backend.host(f), R["recursive_gaussian"], rtol=0, atol=PRODUCT_TOL["recursive"])
    out = backend.empty(SHAPE)
    for var, key in ((4.0, "discrete_gaussian_var4"), (1.0, "discrete_gaussian_var1")):
        ctx.discrete_gaussian(fixed, out, SIZE, SPACING, (var, var, var), 0.01, 32, True)
        np.testing.assert_allclose(backend.host(out), R[key], rtol=0, atol=PRODUCT_TOL["image"])
    ctx.resample(moving, g, g, out, field=field, interp=_lib.INTERP_LINEAR, default_value=-1000.0)
    np.testing.assert_allclose(backend.host(out), R["resample_linear"], rtol=0, atol=PRODUCT_TOL["image"])
    m = backend.empty(SHAPE, np.uint8)
    ctx.resample(mask, g, g, m, field=field, interp=_lib.INTERP_NEAREST, default_value=0, u8=True)
    np.testing.assert_array_equal(backend.host(m), R["resample_nearest"])                  # masks: bit-exact
    ctx.distance_map(mask, g, out, signed=True)
    np.testing.assert_allclose(backend.host(out), R["distance_map_signed"], rtol=PRODUCT_TOL["distance_rel"], atol=PRODUCT_TOL["distance_abs"])
    ctx.label_contour(mask, SIZE, m)
    np.testing.assert_array_equal(backend.host(m), R["label_contour"])


# ---- round 4: the remaining SURVEY 8 rows (tools/sitk_vectors.py::emit_round4).  A key the emitting library could not
# produce is absent (listed in meta_missing) and its check is skipped; with the real SimpleITK every key is there.
ORACLE_TOL.update({"weight_rel": 2e-5, "prob": 2e-6, "metric_rel": 1e-6, "fd_rel": 3e-2})
PRODUCT_TOL.update({"weight_rel": 5e-5, "prob": 5e-6, "metric_rel": 2e-5, "fd_rel": 3e-2, "corner_mm": 0.5})


def _metric_index_map(R):
    """tools/sitk_vectors.py's affine map (physical, about the image centre) as the index-space map the metric kernels take:
    virtual lattice = the fixed grid; moving index = S_m^-1 (A (o_f + S_f v - c) + c + t - o_m)."""
    from tools import sitk_vectors as sv

    A, t, c = np.array(sv.AFFINE_A), np.array(sv.AFFINE_T), np.array(sv.image_centre())
    sp, org = np.array(SPACING), np.array(ORIGIN)
    Am = (A * sp[None, :]) / sp[:, None]
    bm = (A @ (org - c) + c + t - org) / sp
    return A, c, sp, org, Am, bm


def _gradient_wrt_sitk_parameters(r, R):
    """d(mean squared difference) / d(AffineTransform parameters: matrix row-major, translation) from the kernels' sums
    d/dAm, d/dbm (index space) by the chain rule through _metric_index_map."""
    A, c, sp, org, Am, bm = _metric_index_map(R)
    n = r[1]
    dAm, dbm = np.asarray(r[2:11]).reshape(3, 3) / n, np.asarray(r[11:14]) / n
    gA = dAm * sp[None, :] / sp[:, None] + np.outer(dbm / sp, org - c)
    return np.concatenate([gA.ravel(), dbm / sp])


def _atlas_vols(R):
    from tools import sitk_vectors as sv

    return sv.atlas_inputs(R["fixed"], R["moving"], R["mask"])


def check_oracle_against_round4_vectors(R):
    from oracle import linear_oracle
    from tools import sitk_vectors as sv

    vf, vm, vk = O.Vol(R["fixed"], SPACING, ORIGIN), O.Vol(R["moving"], SPACING, ORIGIN), O.Vol(R["mask"], SPACING, ORIGIN)
    checked = []
    if "pyramid_level" in R.files:          # a4, registration/utils.py:216-267
        lvl = O.smooth_and_resample(vf, shrink_factor=sv.PYRAMID_SHRINK, smoothing_sigma=sv.PYRAMID_SIGMA_MM)
        np.testing.assert_allclose(lvl.spacing, R["pyramid_level_spacing"], rtol=1e-12)
        np.testing.assert_allclose(lvl.arr, R["pyramid_level"], rtol=0, atol=ORACLE_TOL["image"])
        checked.append("pyramid_level")
    if "weight_local" in R.files:           # a8, fusion.py:148-190
        np.testing.assert_allclose(O.compute_weight_map(vf, vm, "local").arr, R["weight_local"], rtol=ORACLE_TOL["weight_rel"])
        np.testing.assert_allclose(O.compute_weight_map(vf, vm, "block", dict(sv.BLOCK_PARAMS)).arr, R["weight_block"], rtol=ORACLE_TOL["weight_rel"])
        checked.append("weight maps")
    if "fused_probability" in R.files:      # a9 + a10, fusion.py:263-328
        aset = {}
        for i, (m, l) in enumerate(_atlas_vols(R)):
            aset[str(i)] = {"DIR": {"Weight Map": O.compute_weight_map(vf, O.Vol(m, SPACING, ORIGIN), "local"), "S": O.Vol(l, SPACING, ORIGIN)}}
        p = O.combine_labels(aset, "S")["S"]
        np.testing.assert_allclose(p.arr, R["fused_probability"], rtol=0, atol=ORACLE_TOL["prob"])
        np.testing.assert_array_equal(O.process_probability_image(p, 0.5).arr, R["fused_mask"])
        checked.append("fusion chain")
    if "dilate_ball_221" in R.files:        # f2 / f4, registration/utils.py:328-329
        np.testing.assert_array_equal(O.binary_dilate_ball(vk, (2, 2, 1)).arr, R["dilate_ball_221"])
        np.testing.assert_array_equal(O.binary_closing_ball(O.Vol(NOTCHED(R["mask"]), SPACING, ORIGIN), (2, 1, 0)).arr, R["close_ball_210"])
        checked.append("ball morphology")
    if "fillhole" in R.files:               # f1, fusion.py:308-311
        filled = O.binary_fillhole(O.Vol(sv.cavity_mask(R["mask"]), SPACING, ORIGIN))
        np.testing.assert_array_equal(filled.arr, R["fillhole"])
        np.testing.assert_array_equal(O.connected_component(filled).arr, R["fillhole_component"])
        assert R["fillhole"].sum() > sv.cavity_mask(R["mask"]).sum() and R["fillhole_component"].max() == 2
        checked.append("fill-hole / components")
    if "linear_metric_value" in R.files:    # a7, registration/linear.py:133-153
        A, c, sp, org, Am, bm = _metric_index_map(R)
        r = linear_oracle.meansq_affine(R["fixed"], R["moving"], np.eye(3), np.zeros(3), Am, bm, SIZE, 1)
        np.testing.assert_allclose(r[0] / r[1], float(R["linear_metric_value"]), rtol=ORACLE_TOL["metric_rel"])
        r = linear_oracle.meansq_affine(R["fixed"], R["moving"], np.eye(3), np.zeros(3), Am, bm, SIZE, 1, fixed_mask=R["mask"])
        np.testing.assert_allclose(r[0] / r[1], float(R["linear_metric_masked_value"]), rtol=ORACLE_TOL["metric_rel"])
        g, fd = _gradient_wrt_sitk_parameters(r, R), R["linear_metric_masked_fd_gradient"]
        np.testing.assert_allclose(g, fd, rtol=ORACLE_TOL["fd_rel"], atol=ORACLE_TOL["fd_rel"] * np.abs(fd).max())
        checked.append("linear metric")
    checked += _check_trajectories(R, None)
    return checked


def _check_trajectories(R, pa):
    """Round 6: SimpleITK's optimiser trajectories (tools/sitk_vectors.py::trajectories) against the oracle's registration
    (pa None) or the product's linear_registration: per level the metric value of every iteration the reference reports, the
    final parameters and where the corners land.  Oracle: values 1e-5 relative over the first level (separate fp64
    implementations; a golden-section tie may move a later iteration), corners 0.02 mm.  Product: test_linear_oracle.py's bounds."""
    from oracle import linear_oracle
    from tools import sitk_vectors as sv

    done = []
    for method, optimiser in sv.TRAJECTORY_CASES:
        key = f"linear_trajectory_{method}_{optimiser}"
        if key + "_values" not in R.files:
            continue
        ref = R[key + "_values"]
        kw = sv.TRAJECTORY_KW
        if pa is None:
            got = linear_oracle.registration(O.Vol(R["fixed"], SPACING, ORIGIN), O.Vol(R["moving"], SPACING, ORIGIN), method, optimiser,
                                             kw["shrink_factors"], kw["smooth_sigmas"], kw["sampling_rate"], kw["number_of_iterations"])
            levels = [lv["values"] for lv in got["levels"]]
            A, off = got["matrix_offset"]
            params, tol_first, tol_corner = got["parameters"], 1e-5, 0.02
        else:
            img = lambda a: pa.image_from_array(a, SPACING, ORIGIN)  # noqa: E731
            _, tfm = pa.registration.linear_registration(img(R["fixed"]), img(R["moving"]), reg_method=method, optimiser=optimiser, **kw)
            levels = [lv["values"] for lv in pa.registration.linear_registration.last_levels]
            A, off = tfm.matrix_offset()
            params, tol_first, tol_corner = np.asarray(tfm.transforms[1].GetParameters()), 2e-4, 0.05
        for level in sorted({int(v) for v in ref[:, 0]}):
            want = ref[ref[:, 0] == level][:, 2]
            have = np.asarray(levels[level][:len(want)])
            assert len(levels[level]) in (len(want), len(want) + 1), (key, level, len(levels[level]), len(want))
            rel = np.abs(have - want[:len(have)]) / np.maximum(np.abs(want[:len(have)]), 1e-12)
            # (later levels: a golden-section tie moves an iteration's learning rate by a bracket step and the flat landscape
            # there shows it as a few per cent of a small value -- 3.9 % measured on MI355X; the corners below are the judge)
            assert (rel[:5].max() <= tol_first if level == 0 else True) and rel.max() <= 0.1, (key, level, rel)
        corners = np.array([[ORIGIN[k] + (SHAPE[2 - k] - 1) * SPACING[k] * ((c >> k) & 1) for k in range(3)] for c in range(8)])
        assert np.abs(corners @ np.asarray(A).T + off - R[key + "_corners"]).max() <= tol_corner, key
        # (parameters: translations in mm to the corners' tolerance, matrix / versor entries to 1e-2)
        assert np.abs(np.asarray(params) - R[key + "_parameters"]).max() <= max(tol_corner, 1e-2 * np.abs(R[key + "_parameters"]).max()), key
        done.append(key)
    return done


def check_product_kernels_against_round4_vectors(R, backend):
    from tools import sitk_vectors as sv

    ctx, checked = backend.ctx, []
    if "dilate_ball_221" in R.files:
        for key, radius, op, src in (("dilate_ball_221", (2, 2, 1), 0, R["mask"]), ("close_ball_210", (2, 1, 0), 2, NOTCHED(R["mask"]))):
            out = backend.empty(SHAPE, np.uint8)
            ctx.binary_morph_ball(backend.dev(src), SIZE, radius, op, out)
            np.testing.assert_array_equal(backend.host(out), R[key])
        checked.append("ball morphology")
    if "linear_metric_value" in R.files:
        A, c, sp, org, Am, bm = _metric_index_map(R)
        r = np.array(ctx.meansq_affine(backend.dev(R["fixed"]), SIZE, backend.dev(R["moving"]), SIZE, np.eye(3).ravel(), np.zeros(3), Am.ravel(), bm,
                                       SIZE, 1))
        np.testing.assert_allclose(r[0] / r[1], float(R["linear_metric_value"]), rtol=PRODUCT_TOL["metric_rel"])
        r = np.array(ctx.meansq_affine(backend.dev(R["fixed"]), SIZE, backend.dev(R["moving"]), SIZE, np.eye(3).ravel(), np.zeros(3), Am.ravel(), bm,
                                       SIZE, 1, fixed_mask=backend.dev(R["mask"])))
        np.testing.assert_allclose(r[0] / r[1], float(R["linear_metric_masked_value"]), rtol=PRODUCT_TOL["metric_rel"])
        g, fd = _gradient_wrt_sitk_parameters(r, R), R["linear_metric_masked_fd_gradient"]
        np.testing.assert_allclose(g, fd, rtol=PRODUCT_TOL["fd_rel"], atol=PRODUCT_TOL["fd_rel"] * np.abs(fd).max())
        checked.append("linear metric")
    return checked


def check_product_api_against_round4_vectors(R, pa):
    """The drop-in functions (platipy_amd.*) on the stages whose reference counterpart is a Python function over several sitk calls."""
    from tools import sitk_vectors as sv

    img = lambda a: pa.image_from_array(a, SPACING, ORIGIN)   # noqa: E731
    F, M = img(R["fixed"]), img(R["moving"])
    checked = []
    if "pyramid_level" in R.files:
        lvl = pa.registration.smooth_and_resample(F, shrink_factor=sv.PYRAMID_SHRINK, smoothing_sigma=sv.PYRAMID_SIGMA_MM)
        np.testing.assert_allclose(lvl.GetSpacing(), R["pyramid_level_spacing"], rtol=1e-12)
        np.testing.assert_allclose(lvl.numpy(), R["pyramid_level"], rtol=0, atol=PRODUCT_TOL["image"])
        checked.append("pyramid_level")
    if "weight_local" in R.files:
        np.testing.assert_allclose(pa.label.compute_weight_map(F, M, vote_type="local").numpy(), R["weight_local"], rtol=PRODUCT_TOL["weight_rel"])
        np.testing.assert_allclose(pa.label.compute_weight_map(F, M, vote_type="block", vote_params=dict(sv.BLOCK_PARAMS, normalise=False)).numpy(),
                                   R["weight_block"], rtol=PRODUCT_TOL["weight_rel"])
        checked.append("weight maps")
    if "fused_probability" in R.files:
        aset = {}
        for i, (m, l) in enumerate(_atlas_vols(R)):
            aset[str(i)] = {"DIR": {"Weight Map": pa.label.compute_weight_map(F, img(m), vote_type="local"), "S": img(l)}}
        p = pa.label.combine_labels(aset, "S")["S"]
        np.testing.assert_allclose(p.numpy(), R["fused_probability"], rtol=0, atol=PRODUCT_TOL["prob"])
        np.testing.assert_array_equal(pa.label.process_probability_image(p, 0.5).numpy(), R["fused_mask"])
        checked.append("fusion chain")
    if "fillhole" in R.files:     # the product fuses fill-hole + components + largest (pp_cc.hip): compare with the reference's largest component
        lab = R["fillhole_component"]
        counts = np.bincount(lab.ravel())
        counts[0] = 0
        want = (lab == int(np.argmax(counts))).astype(np.uint8)
        got = pa.label.process_probability_image(img(sv.cavity_mask(R["mask"]).astype(np.float32)), 0.5).numpy()
        np.testing.assert_array_equal(got, want)
        checked.append("fill-hole / components")
    if "linear_similarity_corners" in R.files:
        corners = np.array([[ORIGIN[k] + (SHAPE[2 - k] - 1) * SPACING[k] * ((c >> k) & 1) for k in range(3)] for c in range(8)])
        # the reference's call runs ITK's metric: seeded sample jitter (seed 42) and the filtered gradient image -- the product's
        # itk_sampling=True (round 5) -- so that is the run the reference's corners are held against; the default (lattice
        # samples, analytic gradient) must land there too.  Compared by where the corners land, not by trajectory.
        for kw in (dict(itk_sampling=True), dict()):
            _, tfm = pa.registration.linear_registration(F, M, **sv.LINEAR_KW, **kw)
            A, off = tfm.matrix_offset()
            got = corners @ A.T + off
            assert np.abs(got - R["linear_similarity_corners"]).max() <= PRODUCT_TOL["corner_mm"], kw
        checked.append("linear similarity registration")
    checked += _check_trajectories(R, pa)
    return checked


@pytest.mark.parametrize("path", REFERENCE_FILES or [None], ids=lambda p: os.path.basename(p) if p else "none-committed")
def test_product_api_is_pinned_by_simpleitk_vectors(path, host_api):
    if path is None:
        pytest.skip(NO_REFERENCE)
    check_product_api_against_round4_vectors(np.load(path), host_api)


@pytest.mark.parametrize("path", REFERENCE_FILES or [None], ids=lambda p: os.path.basename(p) if p else "none-committed")
def test_oracle_is_pinned_by_simpleitk_vectors(path):
    if path is None:
        pytest.skip(NO_REFERENCE)
    from tools.sitk_vectors import is_reference

    R = np.load(path)
    assert is_reference(R), f"{path} was not written by SimpleITK (generator {R['meta_generator']}, version {R['meta_sitk_version']})"
    check_oracle_against_reference(R)
    assert len(R["meta_missing"]) == 0, f"the reference file lacks stages: {list(R['meta_missing'])}"
    assert len(check_oracle_against_round4_vectors(R)) == 6


@pytest.mark.parametrize("path", REFERENCE_FILES or [None], ids=lambda p: os.path.basename(p) if p else "none-committed")
def test_product_is_pinned_by_simpleitk_vectors(path, backend):
    if path is None:
        pytest.skip(NO_REFERENCE)
    check_product_against_reference(np.load(path), backend)
    check_product_kernels_against_round4_vectors(np.load(path), backend)


@pytest.fixture
def double_vectors(tmp_path, monkeypatch):
    """A vectors file made by the SAME emit code through tests/sitk_double (the oracle behind sitk's API): plumbing only."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(os.path.join(root, "tests", "sitk_double"))
    sys.modules.pop("SimpleITK", None)
    import SimpleITK

    from tools import sitk_vectors

    assert SimpleITK.__version__ == "test-double"
    dest = str(tmp_path / "sitk_test-double.npz")
    sitk_vectors.emit(SimpleITK, dest, generator="test-double")
    sys.modules.pop("SimpleITK", None)
    R = np.load(dest)
    assert not sitk_vectors.is_reference(R)          # a double's file can never pass for the reference's
    return R


def test_reference_vector_path_runs_end_to_end_with_the_double(double_vectors, backend):
    check_oracle_against_reference(double_vectors)
    check_product_against_reference(double_vectors, backend)
    # round 4: every remaining SURVEY 8 row has its key; the double cannot run ITK's optimiser, which it says in meta_missing
    missing = [str(m) for m in double_vectors["meta_missing"]]
    assert len(missing) == 2 and missing[0].startswith("linear similarity registration") and missing[1].startswith(
        "linear optimiser trajectories"), missing
    assert check_oracle_against_round4_vectors(double_vectors) == ["pyramid_level", "weight maps", "fusion chain", "ball morphology",
                                                                   "fill-hole / components", "linear metric"]
    assert check_product_kernels_against_round4_vectors(double_vectors, backend) == ["ball morphology", "linear metric"]


def test_reference_vector_path_of_the_drop_in_functions_with_the_double(double_vectors, host_api):
    assert check_product_api_against_round4_vectors(double_vectors, host_api) == ["pyramid_level", "weight maps", "fusion chain",
                                                                                   "fill-hole / components"]


def test_emit_command_line_writes_a_file_the_tests_would_pick_up(tmp_path):
    """`tools/compare_with_sitk.py --emit PATH` (the hand-off command) end to end, with the double on PYTHONPATH."""
    import subprocess
    import sys

    from tools.sitk_vectors import is_reference

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dest = tmp_path / "sitk_cli.npz"
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "tests", "sitk_double") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "compare_with_sitk.py"), "--emit", str(dest)], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    R = np.load(dest)
    assert not is_reference(R) and {"execute_1it", "execute_halt_stats", "label_contour", "resample_nearest"} <= set(R.files)
    check_oracle_against_reference(R)


def test_trajectory_checker_runs_on_vectors_of_the_right_shape(tmp_path, host_api):
    """The round-6 trajectory check (_check_trajectories) has no SimpleITK file to run on yet: feed it a file of the emitted
    SHAPE -- [level, iteration, value] rows, parameters, corners -- written from the oracle's own registration of the seeded pair
    (NOT a reference: it only shows that the checker's indexing, level grouping and tolerances execute, for the oracle and for
    the product), so that the day a real file arrives the check cannot fail on its own plumbing."""
    from oracle import linear_oracle
    from tools import sitk_vectors as sv

    fixed, moving, _, _ = sv.inputs()
    vectors = {"fixed": fixed, "moving": moving}
    kw = sv.TRAJECTORY_KW
    corners = np.array([[ORIGIN[k] + (SHAPE[2 - k] - 1) * SPACING[k] * ((c >> k) & 1) for k in range(3)] for c in range(8)])
    for method, optimiser in sv.TRAJECTORY_CASES[:2]:
        got = linear_oracle.registration(O.Vol(fixed, SPACING, ORIGIN), O.Vol(moving, SPACING, ORIGIN), method, optimiser, kw["shrink_factors"],
                                         kw["smooth_sigmas"], kw["sampling_rate"], kw["number_of_iterations"])
        key = f"linear_trajectory_{method}_{optimiser}"
        vectors[key + "_values"] = np.array([[lv, it, v] for lv, rec in enumerate(got["levels"]) for it, v in enumerate(rec["values"])])
        vectors[key + "_parameters"] = got["parameters"]
        A, off = got["matrix_offset"]
        vectors[key + "_corners"] = corners @ A.T + off
    path = str(tmp_path / "shape_only.npz")
    np.savez(path, **vectors)
    R = np.load(path)
    assert _check_trajectories(R, None) == [f"linear_trajectory_{m}_{o}" for m, o in sv.TRAJECTORY_CASES[:2]]
    assert _check_trajectories(R, host_api) == [f"linear_trajectory_{m}_{o}" for m, o in sv.TRAJECTORY_CASES[:2]]
