"""oracle/linear_oracle.py -- TEST INFRASTRUCTURE ONLY (see oracle/pp_oracle.h; PARITY UNPINNED).

Vectorised fp64 restatement of the mean-squares metric evaluation the product computes in
pp_meansq_affine_f32: itk::MeanSquaresImageToImageMetricv4 over the regularly sampled virtual domain
(reference call sites platipy/imaging/registration/linear.py:141-153,238), with the gradient of the
trilinear interpolant as the moving-image gradient."""
import numpy as np


def _sample(img, c):
    """img [Z,Y,X]; c [n,3] continuous indices (x,y,z).  -> inside mask, value, gradient wrt index (n,3)."""
    nz, ny, nx = img.shape
    n = np.array([nx, ny, nz])
    inside = np.all((c >= -0.5) & (c < n[None, :] - 0.5), axis=1)
    cc = np.where(inside[:, None], c, 0.0)
    fl = np.floor(cc)
    b = fl.astype(np.int64)
    i0 = np.maximum(b, 0)
    i1 = np.minimum(i0 + 1, n[None, :] - 1)
    w = np.where(b < 0, 0.0, cc - fl)
    a = img.astype(np.float64)

    def at(ix, iy, iz):
        return a[iz, iy, ix]

    x0, y0, z0, x1, y1, z1 = i0[:, 0], i0[:, 1], i0[:, 2], i1[:, 0], i1[:, 1], i1[:, 2]
    a000, a100 = at(x0, y0, z0), at(x1, y0, z0)
    a010, a110 = at(x0, y1, z0), at(x1, y1, z0)
    a001, a101 = at(x0, y0, z1), at(x1, y0, z1)
    a011, a111 = at(x0, y1, z1), at(x1, y1, z1)
    wx, wy, wz = w[:, 0], w[:, 1], w[:, 2]
    v00, v10 = a000 + (a100 - a000) * wx, a010 + (a110 - a010) * wx
    v01, v11 = a001 + (a101 - a001) * wx, a011 + (a111 - a011) * wx
    v0, v1 = v00 + (v10 - v00) * wy, v01 + (v11 - v01) * wy
    val = v0 + (v1 - v0) * wz
    gx0 = (a100 - a000) + ((a110 - a010) - (a100 - a000)) * wy
    gx1 = (a101 - a001) + ((a111 - a011) - (a101 - a001)) * wy
    g = np.stack([gx0 + (gx1 - gx0) * wz, (v10 - v00) + ((v11 - v01) - (v10 - v00)) * wz, v1 - v0], axis=1)
    return inside, val, g


def _nn_mask(mask, c):
    q = np.floor(c + 0.5).astype(np.int64)
    return mask[q[:, 2], q[:, 1], q[:, 0]] != 0


def meansq_affine(fixed, moving, Af, bf, Am, bm, vsize, stride, fixed_mask=None, moving_mask=None):
    """-> 14 floats: sum (f-m)^2, count, d/dAm (row-major 9), d/dbm (3)."""
    Af, Am = np.asarray(Af, dtype=np.float64).reshape(3, 3), np.asarray(Am, dtype=np.float64).reshape(3, 3)
    bf, bm = np.asarray(bf, dtype=np.float64), np.asarray(bm, dtype=np.float64)
    nv = int(vsize[0]) * int(vsize[1]) * int(vsize[2])
    lin = np.arange(0, nv, int(stride), dtype=np.int64)
    v = np.stack([lin % vsize[0], (lin // vsize[0]) % vsize[1], lin // (vsize[0] * vsize[1])], axis=1).astype(np.float64)
    cf, cm = v @ Af.T + bf, v @ Am.T + bm
    inf_, fval, _ = _sample(np.asarray(fixed), cf)
    inm, mval, g = _sample(np.asarray(moving), cm)
    ok = inf_ & inm
    if fixed_mask is not None:
        ok &= np.where(inf_, _nn_mask(np.asarray(fixed_mask), np.where(inf_[:, None], cf, 0.0)), False)
    if moving_mask is not None:
        ok &= np.where(inm, _nn_mask(np.asarray(moving_mask), np.where(inm[:, None], cm, 0.0)), False)
    diff = np.where(ok, fval - mval, 0.0)
    s = (-2.0 * diff)[:, None] * np.where(ok[:, None], g, 0.0)
    out = np.zeros(14)
    out[0] = float((diff * diff).sum())
    out[1] = float(ok.sum())
    out[2:11] = (s[:, :, None] * v[:, None, :]).sum(0).ravel()
    out[11:14] = s.sum(0)
    return out


def corr_moments_affine(fixed, moving, Af, bf, Am, bm, vsize, stride, fixed_mask=None, moving_mask=None):
    """-> the 42 raw moments pp_corr_moments_affine_f32 accumulates (layout in include/platipy_amd.h)."""
    Af, Am = np.asarray(Af, dtype=np.float64).reshape(3, 3), np.asarray(Am, dtype=np.float64).reshape(3, 3)
    bf, bm = np.asarray(bf, dtype=np.float64), np.asarray(bm, dtype=np.float64)
    nv = int(vsize[0]) * int(vsize[1]) * int(vsize[2])
    lin = np.arange(0, nv, int(stride), dtype=np.int64)
    v = np.stack([lin % vsize[0], (lin // vsize[0]) % vsize[1], lin // (vsize[0] * vsize[1])], axis=1).astype(np.float64)
    cf, cm = v @ Af.T + bf, v @ Am.T + bm
    inf_, fval, _ = _sample(np.asarray(fixed), cf)
    inm, mval, g = _sample(np.asarray(moving), cm)
    ok = inf_ & inm
    if fixed_mask is not None:
        ok &= np.where(inf_, _nn_mask(np.asarray(fixed_mask), np.where(inf_[:, None], cf, 0.0)), False)
    if moving_mask is not None:
        ok &= np.where(inm, _nn_mask(np.asarray(moving_mask), np.where(inm[:, None], cm, 0.0)), False)
    f, m, g, v = fval[ok], mval[ok], g[ok], v[ok]
    terms = np.concatenate([(g[:, :, None] * v[:, None, :]).reshape(len(f), 9), g], axis=1)     # [n, 12]
    out = np.zeros(42)
    out[0:6] = [len(f), f.sum(), m.sum(), (f * f).sum(), (m * m).sum(), (f * m).sum()]
    out[6:18] = terms.sum(0)
    out[18:30] = (f[:, None] * terms).sum(0)
    out[30:42] = (m[:, None] * terms).sum(0)
    return out


# --------------------------------------------------------------------------------------
# mutual information (reference linear.py:145-148: SetMetricAsMattesMutualInformation / ...JointHistogramMutualInformation)


def _bspline3(u):
    a = np.abs(u)
    return np.where(a < 1.0, (4.0 - 6.0 * a * a + 3.0 * a ** 3) / 6.0, np.where(a < 2.0, (2.0 - a) ** 3 / 6.0, 0.0))


def _bspline3_deriv(u):
    a, sg = np.abs(u), np.where(u < 0.0, -1.0, 1.0)
    return np.where(a < 1.0, sg * (-2.0 * a + 1.5 * a * a), np.where(a < 2.0, sg * (-0.5 * (2.0 - a) ** 2), 0.0))


def _mi_samples(fixed, moving, Af, bf, Am, bm, vsize, stride, fixed_mask, moving_mask):
    Af, Am = np.asarray(Af, dtype=np.float64).reshape(3, 3), np.asarray(Am, dtype=np.float64).reshape(3, 3)
    bf, bm = np.asarray(bf, dtype=np.float64), np.asarray(bm, dtype=np.float64)
    nv = int(vsize[0]) * int(vsize[1]) * int(vsize[2])
    lin = np.arange(0, nv, int(stride), dtype=np.int64)
    v = np.stack([lin % vsize[0], (lin // vsize[0]) % vsize[1], lin // (vsize[0] * vsize[1])], axis=1).astype(np.float64)
    cf, cm = v @ Af.T + bf, v @ Am.T + bm
    inf_, fval, _ = _sample(np.asarray(fixed), cf)
    inm, mval, g = _sample(np.asarray(moving), cm)
    ok = inf_ & inm
    if fixed_mask is not None:
        ok &= np.where(inf_, _nn_mask(np.asarray(fixed_mask), np.where(inf_[:, None], cf, 0.0)), False)
    if moving_mask is not None:
        ok &= np.where(inm, _nn_mask(np.asarray(moving_mask), np.where(inm[:, None], cm, 0.0)), False)
    # the kernel interpolates in fp32
    return v[ok], fval[ok].astype(np.float32).astype(np.float64), mval[ok].astype(np.float32).astype(np.float64), g[ok]


def _mi_bins(val, width, norm_min, lo, hi):
    term = val / width - norm_min
    return np.clip(np.floor(term).astype(np.int64), lo, hi), term


def mi_histogram(fixed, moving, Af, bf, Am, bm, vsize, stride, bins, fixed_mask=None, moving_mask=None):
    """bins: dict(nbins, kernel (0 Mattes / 1 joint), f_bin, f_norm_min, m_bin, m_norm_min) -> (hist [nb, nb], count)."""
    v, f, m, _ = _mi_samples(fixed, moving, Af, bf, Am, bm, vsize, stride, fixed_mask, moving_mask)
    nb, pad = int(bins["nbins"]), (2 if bins["kernel"] == 0 else 0)
    fb, _ = _mi_bins(f, bins["f_bin"], bins["f_norm_min"], pad, nb - 1 - pad)
    mb, tm = _mi_bins(m, bins["m_bin"], bins["m_norm_min"], pad, nb - 1 - pad)
    hist = np.zeros((nb, nb))
    if bins["kernel"] == 0:
        for d in (-1, 0, 1, 2):
            np.add.at(hist, (fb, mb + d), _bspline3((mb + d) - tm))
    else:
        np.add.at(hist, (fb, mb), 1.0)
    return hist, float(len(f))


def mi_gradient(fixed, moving, Af, bf, Am, bm, vsize, stride, bins, table, fixed_mask=None, moving_mask=None):
    v, f, m, g = _mi_samples(fixed, moving, Af, bf, Am, bm, vsize, stride, fixed_mask, moving_mask)
    nb, pad = int(bins["nbins"]), (2 if bins["kernel"] == 0 else 0)
    tab = np.asarray(table, dtype=np.float64).astype(np.float32).astype(np.float64)
    fb, _ = _mi_bins(f, bins["f_bin"], bins["f_norm_min"], pad, nb - 1 - pad)
    mb, tm = _mi_bins(m, bins["m_bin"], bins["m_norm_min"], pad, nb - 1 - pad)
    if bins["kernel"] == 0:
        w = sum(_bspline3_deriv((mb + d) - tm) * tab[fb, mb + d] for d in (-1, 0, 1, 2))
    else:
        k0 = np.clip(np.floor(tm - 0.5).astype(np.int64), 0, nb - 2)
        w = tab[fb, k0 + 1] - tab[fb, k0]
    gw = g.astype(np.float32).astype(np.float64) * w[:, None]
    return np.concatenate([(gw[:, :, None] * v[:, None, :]).sum(0).ravel(), gw.sum(0)])
