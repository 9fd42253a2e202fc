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


def meansq_affine(fixed, moving, Af, bf, Am, bm, vsize, stride, fixed_mask=None, moving_mask=None, jitter=None, moving_gradient=None):
    """-> 14 floats: sum (f-m)^2, count, d/dAm (row-major 9), d/dbm (3)."""
    Af, Am = np.asarray(Af, dtype=np.float64).reshape(3, 3), np.asarray(Am, dtype=np.float64).reshape(3, 3)
    bf, bm = np.asarray(bf, dtype=np.float64), np.asarray(bm, dtype=np.float64)
    nv = int(vsize[0]) * int(vsize[1]) * int(vsize[2])
    lin = np.arange(0, nv, int(stride), dtype=np.int64)
    v = np.stack([lin % vsize[0], (lin // vsize[0]) % vsize[1], lin // (vsize[0] * vsize[1])], axis=1).astype(np.float64)
    if jitter is not None:      # ITK's perturbed sample points (itk_regular_jitter), virtual-index units
        v = v + np.asarray(jitter, dtype=np.float64)[:len(v)]
    cf, cm = v @ Af.T + bf, v @ Am.T + bm
    inf_, fval, _ = _sample(np.asarray(fixed), cf)
    inm, mval, g = _sample(np.asarray(moving), cm)
    if moving_gradient is not None:     # ITK's filtered gradient image, linearly interpolated (index units), instead of the interpolant's
        g = np.stack([_sample(np.asarray(moving_gradient[r]), cm)[1] for r in range(3)], axis=1)
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


def corr_moments_affine(fixed, moving, Af, bf, Am, bm, vsize, stride, fixed_mask=None, moving_mask=None, jitter=None, moving_gradient=None):
    """-> the 42 raw moments pp_corr_moments_affine_f32 accumulates (layout in include/platipy_amd.h)."""
    Af, Am = np.asarray(Af, dtype=np.float64).reshape(3, 3), np.asarray(Am, dtype=np.float64).reshape(3, 3)
    bf, bm = np.asarray(bf, dtype=np.float64), np.asarray(bm, dtype=np.float64)
    nv = int(vsize[0]) * int(vsize[1]) * int(vsize[2])
    lin = np.arange(0, nv, int(stride), dtype=np.int64)
    v = np.stack([lin % vsize[0], (lin // vsize[0]) % vsize[1], lin // (vsize[0] * vsize[1])], axis=1).astype(np.float64)
    if jitter is not None:      # ITK's perturbed sample points (itk_regular_jitter), virtual-index units
        v = v + np.asarray(jitter, dtype=np.float64)[:len(v)]
    cf, cm = v @ Af.T + bf, v @ Am.T + bm
    inf_, fval, _ = _sample(np.asarray(fixed), cf)
    inm, mval, g = _sample(np.asarray(moving), cm)
    if moving_gradient is not None:     # ITK's filtered gradient image, linearly interpolated (index units), instead of the interpolant's
        g = np.stack([_sample(np.asarray(moving_gradient[r]), cm)[1] for r in range(3)], axis=1)
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


def _mi_samples(fixed, moving, Af, bf, Am, bm, vsize, stride, fixed_mask, moving_mask, jitter=None, moving_gradient=None):
    Af, Am = np.asarray(Af, dtype=np.float64).reshape(3, 3), np.asarray(Am, dtype=np.float64).reshape(3, 3)
    bf, bm = np.asarray(bf, dtype=np.float64), np.asarray(bm, dtype=np.float64)
    nv = int(vsize[0]) * int(vsize[1]) * int(vsize[2])
    lin = np.arange(0, nv, int(stride), dtype=np.int64)
    v = np.stack([lin % vsize[0], (lin // vsize[0]) % vsize[1], lin // (vsize[0] * vsize[1])], axis=1).astype(np.float64)
    if jitter is not None:      # ITK's perturbed sample points (itk_regular_jitter), virtual-index units
        v = v + np.asarray(jitter, dtype=np.float64)[:len(v)]
    cf, cm = v @ Af.T + bf, v @ Am.T + bm
    inf_, fval, _ = _sample(np.asarray(fixed), cf)
    inm, mval, g = _sample(np.asarray(moving), cm)
    if moving_gradient is not None:     # ITK's filtered gradient image, linearly interpolated (index units), instead of the interpolant's
        g = np.stack([_sample(np.asarray(moving_gradient[r]), cm)[1] for r in range(3)], axis=1)
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


def mi_histogram(fixed, moving, Af, bf, Am, bm, vsize, stride, bins, fixed_mask=None, moving_mask=None, jitter=None):
    """bins: dict(nbins, kernel (0 Mattes / 1 joint), f_bin, f_norm_min, m_bin, m_norm_min) -> (hist [nb, nb], count)."""
    v, f, m, _ = _mi_samples(fixed, moving, Af, bf, Am, bm, vsize, stride, fixed_mask, moving_mask, jitter)
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


def mi_gradient(fixed, moving, Af, bf, Am, bm, vsize, stride, bins, table, fixed_mask=None, moving_mask=None, jitter=None, moving_gradient=None):
    v, f, m, g = _mi_samples(fixed, moving, Af, bf, Am, bm, vsize, stride, fixed_mask, moving_mask, jitter, moving_gradient)
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


# ---- ITK's seeded sample jitter (itk_sampling=True) -----------------------------------------------------------------------
# registration.SetMetricSamplingPercentage(sampling_rate, seed=42) + SetMetricSamplingStrategy(REGULAR)
# (platipy/imaging/registration/linear.py:151-152) -> itk::ImageRegistrationMethodv4::SetMetricSamplePoints: the virtual domain is
# walked in raster order, every ceil(1/rate)-th voxel becomes a sample point, and each coordinate of its PHYSICAL point is
# perturbed by m_Randomizer->GetNormalVariate() * virtualSpacing[d] / 3.  The randomizer is an
# itk::Statistics::MersenneTwisterRandomVariateGenerator seeded once (MetricSamplingReinitializeSeed), so the levels of a
# registration draw from ONE stream, one after the other.  Restated from the published MT19937 reference (Matsumoto &
# Nishimura, init_genrand / genrand_int32) and my reading of ITK 5.3's generator (recollection, PARITY UNPINNED):
#   GetVariateWithOpenRange()      = (int32 + 0.5) / 2^32          GetVariateWithOpenUpperRange() = int32 / 2^32
#   GetNormalVariate(0, 1)         = sqrt(-2 ln(1 - open)) * cos(2 pi upper)      (Box-Muller; `open` is drawn first)
class MersenneTwister:
    """MT19937, 32-bit outputs, seeded by init_genrand(seed) as itk::Statistics::MersenneTwisterRandomVariateGenerator::Initialize."""

    N, M = 624, 397

    def __init__(self, seed):
        st = np.zeros(self.N, dtype=np.uint64)
        st[0] = np.uint64(int(seed) & 0xFFFFFFFF)
        for i in range(1, self.N):
            prev = int(st[i - 1])
            st[i] = np.uint64((1812433253 * (prev ^ (prev >> 30)) + i) & 0xFFFFFFFF)
        self.state = st
        self.pos = self.N

    def _twist(self, lo, hi):
        """state[k] for k in [lo, hi) from state[k], state[k+1] (old) and state[(k+M) % N] (old for k < N-M, new above)."""
        st = self.state
        k = np.arange(lo, hi)
        y = (st[k] & np.uint64(0x80000000)) | (st[(k + 1) % self.N] & np.uint64(0x7FFFFFFF))
        st[k] = st[(k + self.M) % self.N] ^ (y >> np.uint64(1)) ^ np.where(y & np.uint64(1), np.uint64(0x9908B0DF), np.uint64(0))

    def _reload(self):
        n, m = self.N, self.M
        self._twist(0, n - m)                 # reads old state[k + M]
        self._twist(n - m, 2 * (n - m))       # reads state[k - (N - M)], already new
        self._twist(2 * (n - m), n - 1)
        self._twist(n - 1, n)                 # state[N-1] pairs with the NEW state[0]
        self.pos = 0

    def integers(self, count):
        out = np.empty(count, dtype=np.uint64)
        done = 0
        while done < count:
            if self.pos >= self.N:
                self._reload()
            take = min(count - done, self.N - self.pos)
            y = self.state[self.pos:self.pos + take].copy()
            self.pos += take
            y ^= y >> np.uint64(11)
            y ^= (y << np.uint64(7)) & np.uint64(0x9D2C5680)
            y ^= (y << np.uint64(15)) & np.uint64(0xEFC60000)
            y ^= y >> np.uint64(18)
            out[done:done + take] = y & np.uint64(0xFFFFFFFF)
            done += take
        return out

    def normal_variates(self, count):
        """GetNormalVariate(0, 1) x count: two integers each, the radius' first."""
        u = self.integers(2 * count).astype(np.float64)
        r = np.sqrt(-2.0 * np.log(1.0 - (u[0::2] + 0.5) * (1.0 / 4294967296.0)))
        phi = 2.0 * np.pi * (u[1::2] * (1.0 / 4294967296.0))
        return r * np.cos(phi)


def itk_regular_jitter(generator, vsize, stride, vspacing, vdirection=None):
    """Jitter of one level's REGULAR sample points in VIRTUAL-INDEX units, [nsamples, 3] float64: three normal variates per
    sample in raster order (axis 0, 1, 2), each times a third of the virtual spacing of that axis, added to the physical
    point; index = inverse(direction * spacing) applied to that physical offset."""
    nv = int(vsize[0]) * int(vsize[1]) * int(vsize[2])
    nsamp = (nv + int(stride) - 1) // int(stride)
    normals = generator.normal_variates(3 * nsamp).reshape(nsamp, 3)
    sp = np.asarray(vspacing, dtype=np.float64)
    phys = normals * (sp / 3.0)[None, :]
    d = np.eye(3) if vdirection is None else np.asarray(vdirection, dtype=np.float64).reshape(3, 3)
    p2i = np.linalg.inv(d * sp[None, :])
    return phys @ p2i.T
