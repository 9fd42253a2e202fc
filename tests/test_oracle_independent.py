"""Checks of the ORACLE against things that are not the oracle's author's recollection of ITK.

The reference's arithmetic on this path is SimpleITK/ITK, absent here, so `oracle/` is a restatement ("parity
unpinned").  These tests narrow the unpinned surface: every quantity below has a closed form or an
independent implementation (scipy) that does not share code -- or assumptions -- with the oracle:

  * the ESM update on linear-ramp pairs has a closed form; its small-offset limit must be the offset
    itself (Gauss-Newton on a ramp), which pins "J = grad F + grad(M o D)" together with the factor 2, and its maximum
    over the offset must be MaximumUpdateStepLength x RMS spacing, which pins the normaliser;
  * trilinear warp == scipy.ndimage.map_coordinates(order=1) wherever the sample point is inside the buffer;
  * separable smoothing == scipy.ndimage.correlate1d(mode="nearest") with the operator's taps, axis by axis
    (ZeroFluxNeumann = replicate the edge voxel), and the taps == exp(-t) I_k(t) (scipy.special.ive);
  * the recursive Gaussian's impulse response is a unit-gain, symmetric kernel with variance sigma^2 that follows the
    sampled Gaussian within Deriche's published approximation error;
  * resampling between grids == map_coordinates at the mapped continuous indices (order 1 and 0).
What remains recollection (border/sentinel rules of ComputeUpdate, the halt rule, pass orders and intermediate
precisions, SimpleITK's defaults) is listed in DESIGN.md section 3 with the checks tools/compare_with_sitk.py runs.
"""
import numpy as np
import pytest
from scipy import ndimage, special

from oracle import oracle as O


# --------------------------------------------------------------------------------------
# ESM update, closed form on ramps


def _ramp_pair(shape, spacing, grad_mm, delta_mm):
    """F(p) = g . p and W(p) = g . (p - delta) in physical coordinates: s = F - W = g . delta everywhere."""
    nz, ny, nx = shape
    zz, yy, xx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    px, py, pz = xx * spacing[0], yy * spacing[1], zz * spacing[2]
    g = np.asarray(grad_mm, dtype=np.float64)
    dl = np.asarray(delta_mm, dtype=np.float64)
    f = g[0] * px + g[1] * py + g[2] * pz
    w = g[0] * (px - dl[0]) + g[1] * (py - dl[1]) + g[2] * (pz - dl[2])
    return f.astype(np.float32), w.astype(np.float32)


def _closed_form(spacing, grad_mm, delta_mm, step=0.5):
    g = np.asarray(grad_mm, dtype=np.float64)
    s = float(g @ np.asarray(delta_mm, dtype=np.float64))
    J = 2.0 * g                                    # grad F + grad W
    K = np.mean(np.square(spacing)) * step * step  # normaliser
    return 2.0 * s * J / (J @ J + s * s / K), s


@pytest.mark.parametrize("spacing", [(1.0, 1.0, 1.0), (0.8, 1.3, 2.5)])
@pytest.mark.parametrize("grad,delta", [((3.0, 0.0, 0.0), (0.2, 0.0, 0.0)), ((0.0, -2.0, 0.0), (0.0, 0.35, 0.0)),
                                        ((0.0, 0.0, 1.5), (0.0, 0.0, -0.6)), ((2.0, -1.0, 0.5), (0.3, 0.1, -0.2)),
                                        ((4.0, 0.0, 0.0), (3.0, 0.0, 0.0))])
def test_esm_update_closed_form_on_ramps(spacing, grad, delta):
    shape = (9, 10, 11)
    f, w = _ramp_pair(shape, spacing, grad, delta)
    upd, st = O.esm_update(O.Vol(f, spacing), O.Vol(w, spacing))
    want, s = _closed_form(spacing, grad, delta)
    inner = (slice(None), slice(2, -2), slice(2, -2), slice(2, -2))   # away from the one-sided border rules
    got = upd[inner].reshape(3, -1)
    # fp32 images: s and the gradients carry ~1e-6 relative error
    np.testing.assert_allclose(got, np.repeat(want[:, None], got.shape[1], axis=1), rtol=2e-4, atol=2e-6)
    # metric = mean s^2 over every voxel (s is constant), RMS change from the raw update
    np.testing.assert_allclose(st.metric, s * s, rtol=1e-4)
    assert st.n_pixels == f.size


def test_esm_update_small_offset_recovers_the_offset_and_step_is_bounded():
    """Gauss-Newton on a ramp: U -> delta as delta -> 0; and max_delta |U| = MaximumUpdateStepLength * RMS(spacing)."""
    spacing = (0.9, 1.1, 2.0)
    f, w = _ramp_pair((7, 8, 12), spacing, (5.0, 0.0, 0.0), (1e-2, 0.0, 0.0))
    upd, _ = O.esm_update(O.Vol(f, spacing), O.Vol(w, spacing))
    np.testing.assert_allclose(upd[0, 3, 4, 4:8], 1e-2, rtol=2e-3)
    np.testing.assert_allclose(upd[1:, 3, 4, 4:8], 0.0, atol=1e-9)
    bound = 0.5 * np.sqrt(np.mean(np.square(spacing)))
    best = 0.0
    for dl in np.linspace(0.1, 6.0, 60):
        f, w = _ramp_pair((5, 6, 16), spacing, (5.0, 0.0, 0.0), (dl, 0.0, 0.0))
        upd, _ = O.esm_update(O.Vol(f, spacing), O.Vol(w, spacing))
        best = max(best, float(np.abs(upd[0, 2, 3, 6:10]).max()))
        assert np.abs(upd[:, 2, 3, 6:10]).max() <= bound * (1 + 1e-5)
    np.testing.assert_allclose(best, bound, rtol=2e-3)   # attained at |delta| = 2 sqrt(K)


def test_esm_update_is_zero_where_images_agree_or_difference_is_below_threshold():
    f, _ = _ramp_pair((6, 7, 8), (1, 1, 1), (2.0, 1.0, 0.5), (0, 0, 0))
    upd, st = O.esm_update(O.Vol(f, (1, 1, 1)), O.Vol(f.copy(), (1, 1, 1)))
    assert np.all(upd == 0.0) and st.metric == 0.0
    w = f + np.float32(5e-4)    # |s| < IntensityDifferenceThreshold = 1e-3
    upd, st = O.esm_update(O.Vol(f, (1, 1, 1)), O.Vol(w, (1, 1, 1)))
    assert np.all(upd == 0.0)


def test_esm_border_and_sentinel_rules_by_hand():
    """ITK's ComputeUpdate on a 1-D row, worked by hand from the documented rules (this is a restatement check, listed
    as such in DESIGN section 3): fixed gradient = central difference, 0 on the first/last index; warped gradient =
    central difference, one-sided next to the border or a sentinel neighbour, 0 with no usable neighbour; a sentinel
    centre voxel gets no update and is not counted."""
    sent = np.finfo(np.float32).max
    f = np.array([0.0, 1.0, 3.0, 6.0, 10.0, 15.0], dtype=np.float32).reshape(1, 1, 6)
    w = np.array([0.5, 2.0, sent, 5.0, 9.0, 13.0], dtype=np.float32).reshape(1, 1, 6)
    upd, st = O.esm_update(O.Vol(f, (1, 1, 1)), O.Vol(w, (1, 1, 1)))
    K = 0.25

    def u(s, j):
        return 2 * s * j / (j * j + s * s / K)

    want = [
        u(0.0 - 0.5, 0.0 + (2.0 - 0.5)),          # first index: fixed gradient 0, warped one-sided forward
        u(1.0 - 2.0, (3.0 - 0.0) / 2 + (2.0 - 0.5)),  # right neighbour is the sentinel: warped one-sided backward
        0.0,                                       # sentinel centre
        u(6.0 - 5.0, (10.0 - 3.0) / 2 + (9.0 - 5.0)),  # left neighbour is the sentinel: warped one-sided forward
        u(10.0 - 9.0, (15.0 - 6.0) / 2 + (13.0 - 5.0) / 2),
        u(15.0 - 13.0, 0.0 + (13.0 - 9.0)),       # last index
    ]
    np.testing.assert_allclose(upd[0, 0, 0], want, rtol=1e-6, atol=1e-12)
    assert st.n_pixels == 5


# --------------------------------------------------------------------------------------
# warp and resample vs scipy


def _smooth(shape, seed, amp):
    rng = np.random.default_rng(seed)
    return ndimage.gaussian_filter(rng.normal(size=shape), 2.0) * amp


def test_warp_matches_map_coordinates_in_the_interior():
    shape, spacing = (14, 17, 19), (1.2, 0.9, 2.0)
    img = (_smooth(shape, 1, 300.0) + 50.0).astype(np.float32)
    dvf = np.stack([_smooth(shape, 10 + c, 12.0) for c in range(3)])          # mm; components x, y, z
    got = O.warp_image(O.Vol(img, spacing), dvf, edge_value=-777.0).arr
    zz, yy, xx = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
    cz, cy, cx = zz + dvf[2] / spacing[2], yy + dvf[1] / spacing[1], xx + dvf[0] / spacing[0]
    want = ndimage.map_coordinates(img.astype(np.float64), [cz, cy, cx], order=1, mode="nearest")
    inside = (cz >= 0) & (cz <= shape[0] - 1) & (cy >= 0) & (cy <= shape[1] - 1) & (cx >= 0) & (cx <= shape[2] - 1)
    assert inside.mean() > 0.8 and np.abs(dvf).max() > 0.5
    np.testing.assert_allclose(got[inside], want[inside], rtol=1e-5, atol=2e-3)
    # outside the buffer (more than half a voxel beyond the last index): the edge value
    outside = (cz < -0.5) | (cz >= shape[0] - 0.5) | (cy < -0.5) | (cy >= shape[1] - 0.5) | (cx < -0.5) | (cx >= shape[2] - 0.5)
    assert np.all(got[outside] == np.float32(-777.0))


@pytest.mark.parametrize("interp,order", [(O.INTERP_LINEAR, 1), (O.INTERP_NEAREST, 0)])
def test_resample_between_grids_matches_map_coordinates(interp, order):
    shape, spacing, origin = (12, 15, 18), (1.0, 1.5, 2.0), (3.0, -4.0, 10.0)
    img = (_smooth(shape, 3, 200.0)).astype(np.float32)
    oshape, ospacing, oorigin = (9, 11, 13), (1.37, 1.91, 2.63), (4.1, -2.2, 11.3)
    ref = O.Vol(np.zeros(oshape, np.float32), ospacing, oorigin)
    got = O.resample(O.Vol(img, spacing, origin), ref, interp=interp, default_value=-5.0).arr
    zz, yy, xx = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in oshape], indexing="ij")
    cx = (oorigin[0] + xx * ospacing[0] - origin[0]) / spacing[0]
    cy = (oorigin[1] + yy * ospacing[1] - origin[1]) / spacing[1]
    cz = (oorigin[2] + zz * ospacing[2] - origin[2]) / spacing[2]
    inside = (cz >= 0) & (cz <= shape[0] - 1) & (cy >= 0) & (cy <= shape[1] - 1) & (cx >= 0) & (cx <= shape[2] - 1)
    if order == 0:
        # nearest neighbour: avoid exact ties (scipy rounds half to even, ITK half up)
        frac = np.stack([c - np.floor(c) for c in (cz, cy, cx)])
        inside &= np.all(np.abs(frac - 0.5) > 1e-6, axis=0)
    want = ndimage.map_coordinates(img.astype(np.float64), [cz, cy, cx], order=order, mode="nearest")
    assert inside.mean() > 0.5
    np.testing.assert_allclose(got[inside], want[inside], rtol=1e-5, atol=1e-3 if order else 0.0)


# --------------------------------------------------------------------------------------
# FIR smoothing vs scipy.ndimage.correlate1d


def test_gaussian_operator_taps_are_scaled_bessel_functions():
    for var, err in [(1.0, 0.1), (2.25, 0.1), (0.36, 0.1), (1.0, 0.01), (4.0, 0.01), (64.0, 0.01)]:
        taps = O.gaussian_operator(var, err, 1000)
        r = len(taps) // 2
        k = np.arange(-r, r + 1)
        raw = special.ive(np.abs(k), var)          # exp(-t) I_k(t), the discrete Gaussian of Lindeberg
        assert raw.sum() >= 1 - err and (r == 0 or raw[1:-1].sum() < 1 - err)   # truncated where the sum reaches 1 - err
        np.testing.assert_allclose(taps, raw / raw.sum(), rtol=5e-6)              # ITK uses polynomial Bessel fits
        np.testing.assert_allclose(taps.sum(), 1.0, rtol=1e-12)


def test_field_smoothing_is_separable_correlation_with_replicated_edges():
    shape = (9, 12, 14)
    f = np.stack([_smooth(shape, 20 + c, 3.0) for c in range(3)])
    sig = (1.5, 0.9, 1.2)       # voxels, per axis x, y, z
    got = O.smooth_field(f, sig)
    want = f.copy()
    for axis_xyz, s in enumerate(sig):
        taps = O.gaussian_operator(s * s, 0.1, 30)
        want = ndimage.correlate1d(want, taps, axis=3 - axis_xyz, mode="nearest")
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)   # separable passes commute in exact arithmetic


def test_discrete_gaussian_is_separable_correlation_in_physical_units():
    shape, spacing = (10, 12, 16), (1.0, 1.25, 2.5)
    img = (_smooth(shape, 30, 100.0)).astype(np.float32)
    var_mm2 = 4.0
    got = O.discrete_gaussian(O.Vol(img, spacing), var_mm2, max_kernel_width=32, max_error=0.01).arr
    want = img.astype(np.float64)
    for axis_xyz, sp in enumerate(spacing):
        taps = O.gaussian_operator(var_mm2 / (sp * sp), 0.01, 32)
        want = ndimage.correlate1d(want, taps, axis=2 - axis_xyz, mode="nearest")
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-5)     # fp32 intermediate images
    # constants and (away from the edges) linear ramps pass unchanged
    c = O.discrete_gaussian(O.Vol(np.full(shape, 7.5, np.float32), spacing), var_mm2).arr
    np.testing.assert_allclose(c, 7.5, rtol=1e-6)


# --------------------------------------------------------------------------------------
# recursive (Deriche) Gaussian


@pytest.mark.parametrize("sigma", [1.5, 2.5, 4.0])
def test_recursive_gaussian_impulse_response(sigma):
    n = 65
    imp = np.zeros((n, n, n), np.float32)
    imp[n // 2, n // 2, n // 2] = 1.0
    out = O.recursive_gaussian(O.Vol(imp, (1.0, 1.0, 1.0)), [sigma, sigma, sigma]).arr.astype(np.float64)
    x = np.arange(n) - n // 2
    gauss = np.exp(-0.5 * (x / sigma) ** 2) / (sigma * np.sqrt(2 * np.pi))
    np.testing.assert_allclose(out.sum(), 1.0, rtol=5e-5)                           # unit DC gain (3 axes)
    for axis in range(3):
        # the filter is separable: the marginal over the other two axes is the 1-D impulse response along `axis`
        line = out.sum(axis=tuple(a for a in range(3) if a != axis))
        np.testing.assert_allclose(line, line[::-1], atol=5e-6)                     # symmetric
        assert np.abs(line - gauss).max() <= 0.004 * gauss.max()                    # Deriche's 4th-order fit: 0.3 % of the peak
        # its small negative side lobes lower the second moment by 6.2 % at every scale (a property of the published fit)
        np.testing.assert_allclose((line * x * x).sum() / (sigma * sigma), 0.938, atol=0.003)


def test_first_order_recursive_gaussian_is_the_derivative_of_a_gaussian():
    """The FirstOrder directional filter (ITK's second column of Deriche constants, recalled from memory) held to things that do
    not share its code: a ramp of slope s per voxel answers s exactly (the alpha1 normalisation), the impulse response is
    antisymmetric, and it is the analytic derivative of the sampled Gaussian to within Deriche's approximation error (0.3 % of
    the peak); NormalizeAcrossScale multiplies by sigma."""
    n = 64
    for sp, sigma in ((1.0, 2.0), (1.5, 2.5), (2.5, 2.5)):
        ramp = np.zeros((n, 4, 4), np.float32) + (np.arange(n, dtype=np.float32) * 3.0)[:, None, None]
        out = O.recursive_gaussian_pass(O.Vol(ramp, (1.0, 1.0, sp)), 2, sigma, 1, False).arr
        np.testing.assert_allclose(out[24:40, 1, 2], 3.0, rtol=0, atol=3e-6)      # (away from the edge-extended ends)
        imp = np.zeros((n, 4, 4), np.float32)
        imp[32] = 1.0
        r = O.recursive_gaussian_pass(O.Vol(imp, (1.0, 1.0, sp)), 2, sigma, 1, False).arr[:, 0, 0].astype(np.float64)
        assert np.abs(r[33:42] + r[31:22:-1]).max() < 1e-7 and abs(r[32]) < 1e-7
        x, sd = np.arange(n) - 32.0, sigma / sp
        dG = -x / sd ** 2 * np.exp(-0.5 * (x / sd) ** 2) / (sd * np.sqrt(2 * np.pi))
        assert np.abs(r - dG).max() < 4e-3 * np.abs(dG).max()
        scaled = O.recursive_gaussian_pass(O.Vol(imp, (1.0, 1.0, sp)), 2, sigma, 1, True).arr[:, 0, 0]
        np.testing.assert_allclose(scaled, sigma * r, rtol=0, atol=2e-6)
    # the gradient filter of the metric: a linear intensity field a.x + b.y + c.z (mm) has the gradient sigma * (a, b, c) per mm
    zz, yy, xx = np.meshgrid(np.arange(24) * 2.5, np.arange(30) * 1.2, np.arange(32) * 1.5, indexing="ij")
    lin = (2.0 * xx - 3.0 * yy + 0.5 * zz).astype(np.float32)
    g = O.gradient_recursive_gaussian(O.Vol(lin, (1.5, 1.2, 2.5)))
    for c, want in enumerate((2.0, -3.0, 0.5)):
        np.testing.assert_allclose(g[c][9:15, 12:18, 13:19], 2.5 * want, rtol=5e-4, atol=2e-5)   # (the volume's middle: the ends feel the edge extension)
