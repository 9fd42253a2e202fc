"""Iterative atlas removal (platipy_amd.label.run_iar) against the oracle's restatement of one pass
(platipy/imaging/label/iar.py:91-229) and by what it does: a grossly wrong atlas is removed."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import smooth_noise

SHAPE, SPACING, ORIGIN = (24, 40, 48), (1.1, 0.9, 2.0), (5.0, -3.0, 2.0)


def _labels(n):
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in SHAPE], indexing="ij")
    out = []
    for i in range(n):
        r = 13 + 0.5 * smooth_noise(SHAPE, 300 + i, cells=4)
        cx, cy, cz = 24 + 0.3 * (i % 3), 20 - 0.3 * (i % 2), 12
        if i == n - 1:  # the outlier: shifted and shrunk
            cx, cy, r = cx + 7, cy - 5, r - 4
        m = ((xx - cx) * 1.0) ** 2 + ((yy - cy) * 1.0) ** 2 + ((zz - cz) * 1.6) ** 2 <= r ** 2
        out.append(m.astype(np.uint8))
    return out


def test_run_iar_matches_oracle_and_removes_outlier(host_api):
    pa = host_api
    labs = _labels(9)
    w = np.ones(SHAPE, np.float32)
    aset_g, aset_o = {}, {}
    for i, lab in enumerate(labs):
        cid = f"{i:02d}"
        aset_g[cid] = {"DIR": {"Weight Map": pa.image_from_array(w, SPACING, ORIGIN), "HEART": pa.image_from_array(lab, SPACING, ORIGIN)}}
        aset_o[cid] = {"DIR": {"Weight Map": O.Vol(w, SPACING, ORIGIN), "HEART": O.Vol(lab, SPACING, ORIGIN)}}
    # distance samples: same reference contour, same exact EDT
    ref = O.process_probability_image(O.combine_labels(aset_o, "HEART")["HEART"], 0.95)
    want = O.evaluate_distance_to_reference(ref, O.Vol(labs[3], SPACING, ORIGIN), 5)
    got = pa.label.evaluate_distance_to_reference(pa.image_from_array(ref.arr, SPACING, ORIGIN), pa.image_from_array(labs[3], SPACING, ORIGIN), 5)
    assert got.shape == want.shape and got.size > 50
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-5)
    # one pass: Q metrics agree, the outlier has the largest Q and is removed
    kept = pa.label.run_iar(aset_g, "HEART", min_best_atlases=4, single_step=True)
    q_g = pa.label.run_iar.last_q_results
    q_o = O.iar_q_values(aset_o, "HEART")
    assert list(q_g) == list(q_o)
    np.testing.assert_allclose([q_g[k] for k in q_g], [q_o[k] for k in q_o], rtol=1e-3, atol=1e-6)
    assert max(q_g, key=q_g.get) == "08"
    assert "08" not in kept and len(kept) >= 4
    # full recursion terminates and keeps a consistent set
    final = pa.label.run_iar(aset_g, "HEART", min_best_atlases=4)
    assert "08" not in final and set(final) <= set(aset_g)


@pytest.mark.parametrize("n_atlases", [4, 7])
@pytest.mark.parametrize("statistic", ["mad", "std"])
def test_leave_one_out_z_scores_on_the_device_match_numpy(n_atlases, statistic):
    """run_iar's device form of the leave-one-out robust z-scores (iar.py:166-199) is numpy's median / MAD arithmetic:
    even and odd atlas counts, columns with zero spread included."""
    import torch

    from platipy_amd.label import iar

    rng = np.random.default_rng(3)
    samples = rng.gamma(2.0, 1.5, size=(n_atlases, 501)).astype(np.float32)
    samples[:, 7] = 1.25                      # zero spread in one column: the rule substitutes the median / mean spread
    samples[1:, 100] = 0.5
    dev = torch.from_numpy(samples)
    for k in range(n_atlases):
        want = iar._z_scores(samples[k], np.delete(samples, k, axis=0), statistic)
        got = iar._z_scores_device(dev, k, statistic)
        if statistic == "mad":
            np.testing.assert_array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-6)     # mean / std: different summation order


def test_gaussian_curve_is_scipy_s_norm_pdf_bit_for_bit():
    """label/iar.py::gaussian_curve restates a * scipy.stats.norm.pdf(x, loc=m, scale=s) (reference iar.py:55-56) without
    rv_continuous's argument handling: the same floats, NaN for a non-positive scale, so curve_fit walks the same path."""
    from scipy.stats import norm

    from platipy_amd.label.iar import gaussian_curve

    x = (np.linspace(-15, 15, 501)[1:] + np.linspace(-15, 15, 501)[:-1]) / 2.0
    rng = np.random.default_rng(5)
    for _ in range(50):
        a, m, s = rng.uniform(0.1, 3.0), rng.normal(0, 2.0), rng.uniform(0.05, 6.0)
        assert np.array_equal(gaussian_curve(x, a, m, s), a * norm.pdf(x, loc=m, scale=s))
    for s in (0.0, -1.5):
        got, want = gaussian_curve(x, 1.2, 0.3, s), 1.2 * norm.pdf(x, loc=0.3, scale=s)
        assert np.isnan(got).all() and np.isnan(want).all()
