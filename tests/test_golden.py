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
