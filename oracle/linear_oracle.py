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


# =================================================================================================================================
# The registration as a whole (round 6): sitk.ImageRegistrationMethod.Execute as platipy/imaging/registration/linear.py:129-238
# configures it, restated in fp64 numpy from the ITK 5.3 classes it instantiates -- ImageRegistrationMethodv4 (level loop),
# MeanSquaresImageToImageMetricv4 (physical-space formulation, analytic transform Jacobians),
# RegistrationParameterScalesFromPhysicalShift, GradientDescentOptimizerv4 / GradientDescentLineSearchOptimizerv4,
# WindowConvergenceMonitoringFunction, the transforms' UpdateTransformParameters.  TEST INFRASTRUCTURE, PARITY UNPINNED (ITK from
# memory); it shares no code with platipy_amd/registration/linear.py or csrc/pp_linear.hip: the product works in index space with
# central differences of the index map, batches its line-search probes and runs its metric in fp32 kernels; this file works in
# physical space with analytic Jacobians, probes sequentially and accumulates in fp64.


def _quat_mul(a, b):
    """Hamilton product of (v, w) quaternions, vector form: (wa vb + wb va + va x vb, wa wb - va . vb)."""
    va, wa, vb, wb = np.asarray(a[:3]), a[3], np.asarray(b[:3]), b[3]
    return np.concatenate([wa * vb + wb * va + np.cross(va, vb), [wa * wb - va @ vb]])


class OracleTransform:
    """The optimised transform T(x) = M(p) (x - c) + c + t(p) with ITK's parameter layouts, its analytic Jacobian and its
    UpdateTransformParameters.  kinds: translation (t), rigid (versor 3, t 3), similarity (versor 3, t 3, scale), affine (matrix
    row-major 9, t 3)."""

    def __init__(self, kind, center=(0.0, 0.0, 0.0)):
        self.kind = kind
        self.c = np.asarray(center, dtype=np.float64)
        self.p = {"translation": np.zeros(3), "rigid": np.zeros(6), "similarity": np.array([0, 0, 0, 0, 0, 0, 1.0]),
                  "affine": np.concatenate([np.eye(3).ravel(), np.zeros(3)])}[kind].astype(np.float64)

    @staticmethod
    def _rot(v):
        """itk::Versor -> matrix: R x = x + 2 w (v x x) + 2 v x (v x x), w = sqrt(1 - |v|^2)."""
        v = np.asarray(v, dtype=np.float64)
        w = np.sqrt(max(0.0, 1.0 - v @ v))
        K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        return np.eye(3) + 2.0 * w * K + 2.0 * K @ K

    def matrix_translation(self, p=None):
        p = self.p if p is None else p
        if self.kind == "translation":
            return np.eye(3), p[0:3]
        if self.kind == "rigid":
            return self._rot(p[0:3]), p[3:6]
        if self.kind == "similarity":
            return p[6] * self._rot(p[0:3]), p[3:6]
        return p[0:9].reshape(3, 3), p[9:12]

    def apply(self, x, p=None):
        M, t = self.matrix_translation(p)
        return (x - self.c) @ M.T + self.c + t

    def jacobian(self, x):
        """d T(x) / d p  -> [n, 3, n_params] (ComputeJacobianWithRespectToParameters)."""
        d = x - self.c
        n = len(x)
        if self.kind == "translation":
            return np.broadcast_to(np.eye(3), (n, 3, 3)).copy()
        if self.kind == "affine":
            J = np.zeros((n, 3, 12))
            for r in range(3):
                J[:, r, 3 * r:3 * r + 3] = d
                J[:, r, 9 + r] = 1.0
            return J
        v = self.p[0:3]
        w = np.sqrt(max(1e-300, 1.0 - v @ v))
        s = self.p[6] if self.kind == "similarity" else 1.0
        J = np.zeros((n, 3, len(self.p)))
        vxd = np.cross(v, d)
        for i in range(3):
            e = np.zeros(3)
            e[i] = 1.0
            exd = np.cross(e, d)
            J[:, :, i] = s * (2.0 * (-v[i] / w) * vxd + 2.0 * w * exd + 2.0 * (np.cross(e, vxd) + np.cross(v, exd)))
        J[:, 0, 3] = J[:, 1, 4] = J[:, 2, 5] = 1.0
        if self.kind == "similarity":
            J[:, :, 6] = d @ self._rot(v).T
        return J

    def updated(self, update):
        """UpdateTransformParameters(update, 1) -> new parameter vector (self.p untouched)."""
        update = np.asarray(update, dtype=np.float64)
        new = self.p + update
        if self.kind in ("rigid", "similarity"):
            v = self.p[0:3]
            cur = np.concatenate([v, [np.sqrt(max(0.0, 1.0 - v @ v))]])
            angle = float(np.linalg.norm(update[0:3]))
            if angle > 0.0:
                grad_rot = np.concatenate([update[0:3] / angle * np.sin(angle / 2.0), [np.cos(angle / 2.0)]])
            else:
                grad_rot = np.array([0.0, 0.0, 0.0, 1.0])
            q = _quat_mul(cur, grad_rot)
            new[0:3] = q[0:3] if q[3] >= 0.0 else -q[0:3]
        return new


def _window_convergence_itk(energies, window=10):
    """WindowConvergenceMonitoringFunction::GetConvergenceValue (see the note in the product's _window_convergence; written here
    from the filter's per-point accumulation rather than in closed form)."""
    if len(energies) < window:
        return np.inf
    e = np.asarray(energies[-window:], dtype=np.float64)
    total = np.abs(e).sum()
    if total == 0.0:
        return 0.0
    delta, omega = np.zeros(2), np.zeros(2)
    for i in range(window):
        u = i / (window - 1.0)
        if abs(u - 1.0) <= 1e-4:
            u = 1.0 - 1e-4
        bw = np.array([1.0 - u, u])                  # order-1 B-spline weights of the two control points
        w2 = (bw * bw).sum()
        for k in range(2):
            phi = bw[k] * (e[i] / total) / w2
            delta[k] += bw[k] * bw[k] * phi
            omega[k] += bw[k] * bw[k]
    lattice = delta / omega
    return -(lattice[1] - lattice[0])


def _shrunk_virtual_domain(size, spacing, origin, direction, factor):
    """itk::ShrinkImageFilter::GenerateOutputInformation: size floor(n / f) (>= 1), spacing * f, origin moved so that the centre
    of the domain stays where it is."""
    n = np.asarray(size, dtype=np.float64)
    f = np.broadcast_to(np.asarray(factor, dtype=np.float64), (3,))
    out_size = np.maximum(1, np.floor(n / f)).astype(np.int64)
    sp_in = np.asarray(spacing, dtype=np.float64)
    sp_out = sp_in * f
    D = np.asarray(direction, dtype=np.float64).reshape(3, 3)
    centre_in = D @ (sp_in * (n - 1.0) / 2.0)
    centre_out = D @ (sp_out * (out_size - 1.0) / 2.0)
    return out_size, sp_out, np.asarray(origin, dtype=np.float64) + centre_in - centre_out, D


class _PhysicalMeanSquares:
    """MeanSquaresImageToImageMetricv4 -- or, metric="correlation", CorrelationImageToImageMetricv4: value -(sum (f - fbar)(m - mbar))^2
    / (sum (f - fbar)^2 sum (m - mbar)^2) over the valid samples -- over one level's sample point set, in PHYSICAL space."""

    def __init__(self, fixed, moving, points, init_matrix, init_offset, gradient_image, metric="mean_squares"):
        self.metric = metric
        self.f_arr, self.m_arr = np.asarray(fixed.arr, dtype=np.float32), np.asarray(moving.arr, dtype=np.float32)
        self.f_p2i, self.f_o = self._p2i(fixed), np.asarray(fixed.origin, dtype=np.float64)
        self.m_p2i, self.m_o = self._p2i(moving), np.asarray(moving.origin, dtype=np.float64)
        self.m_i2p_T = np.linalg.inv(self.m_p2i).T
        self.points = points
        self.Ai, self.oi = init_matrix, init_offset
        self.gimg = gradient_image              # [3, Z, Y, X] physical gradient (filtered), or None: the interpolant's own
        ok, self.fval, _ = _sample(self.f_arr, (points - self.f_o) @ self.f_p2i.T)
        self.f_ok = ok
        self.evaluations = 0

    @staticmethod
    def _p2i(vol):
        D = np.asarray(vol.direction, dtype=np.float64).reshape(3, 3)
        return np.linalg.inv(D * np.asarray(vol.spacing, dtype=np.float64)[None, :])

    def _moving(self, tfm, p):
        y = tfm.apply(self.points, p) @ self.Ai.T + self.oi
        cm = (y - self.m_o) @ self.m_p2i.T
        ok, mval, g_idx = _sample(self.m_arr, cm)
        return ok & self.f_ok, mval, g_idx, cm

    def _correlation(self, ok, mval):
        f, m = self.fval[ok], mval[ok]
        fc, mc = f - f.mean(), m - m.mean()
        sff, smm, sfm = float((fc * fc).sum()), float((mc * mc).sum()), float((fc * mc).sum())
        return fc, mc, sff, smm, sfm

    def value(self, tfm, p=None):
        self.evaluations += 1
        ok, mval, _, _ = self._moving(tfm, p)
        n = int(ok.sum())
        if n == 0:
            return np.inf
        if self.metric == "correlation":
            _, _, sff, smm, sfm = self._correlation(ok, mval)
            return 0.0 if (sff <= 1e-300 or smm <= 1e-300) else -(sfm * sfm) / (sff * smm)
        d = np.where(ok, self.fval - mval, 0.0)
        return float((d * d).sum() / n)

    def value_and_derivative(self, tfm):
        """-> (value, derivative) with ITK's sign: the optimiser ADDS learning_rate * derivative / scales."""
        self.evaluations += 1
        ok, mval, g_idx, cm = self._moving(tfm, None)
        n = int(ok.sum())
        if n == 0:
            return np.inf, np.zeros(len(tfm.p))
        if self.gimg is not None:
            g_phys = np.stack([_sample(self.gimg[r], cm)[1] for r in range(3)], axis=1)
        else:
            g_phys = g_idx @ self.m_p2i              # d m / d y = (d idx / d y)^T d m / d idx
        d = np.where(ok, self.fval - mval, 0.0)
        J = np.einsum("rc,ncp->nrp", self.Ai, tfm.jacobian(self.points))    # composite: initial's position Jacobian x optimised's
        per_sample = np.einsum("nr,nrp->np", np.where(ok[:, None], g_phys, 0.0), J)
        if self.metric == "correlation":
            # d value / d p with m_s = m(T(x_s; p)):  d sfm = sum fc_s dm_s,  d smm = 2 sum mc_s dm_s  (the means' own derivatives
            # cancel against sum fc = sum mc = 0);  ITK's sign: the optimiser ADDS the returned derivative, so it is minus the gradient
            fc, mc, sff, smm, sfm = self._correlation(ok, mval)
            if sff <= 1e-300 or smm <= 1e-300:
                return 0.0, np.zeros(len(tfm.p))
            dm = per_sample[ok]
            dsfm, dsmm = (fc[:, None] * dm).sum(0), 2.0 * (mc[:, None] * dm).sum(0)
            gradient = -(2.0 * sfm / (sff * smm) * dsfm - (sfm * sfm) / (sff * smm * smm) * dsmm)
            return -(sfm * sfm) / (sff * smm), -gradient
        return float((d * d).sum() / n), (2.0 * d[:, None] * per_sample).sum(0) / n


def _corner_points(vsize, vspacing, vorigin, vdir):
    n = np.asarray(vsize, dtype=np.float64) - 1.0
    idx = np.array([[i, j, k] for k in (0.0, n[2]) for j in (0.0, n[1]) for i in (0.0, n[0])])
    return vorigin[None, :] + (idx * vspacing[None, :]) @ vdir.T


def _max_shift(tfm, corners, Ai, delta):
    """ScalesFromShiftBase::ComputeMaximumVoxelShift for the physical-shift estimator: largest displacement of a corner of the
    virtual domain, through the composite moving transform, when the parameters are UPDATED by delta."""
    old = tfm.apply(corners) @ Ai.T
    new = tfm.apply(corners, tfm.updated(delta)) @ Ai.T
    return float(np.sqrt(((new - old) ** 2).sum(1)).max())


def _estimate_scales(tfm, corners, Ai, variation=0.01):
    n = len(tfm.p)
    shifts = np.zeros(n)
    for i in range(n):
        d = np.zeros(n)
        d[i] = variation
        shifts[i] = _max_shift(tfm, corners, Ai, d)
    eps = np.finfo(np.float64).eps
    nonzero = shifts[shifts > eps]
    if nonzero.size == 0:
        return np.ones(n)
    smallest = nonzero.min()
    scales = np.where(shifts <= eps, smallest * smallest, shifts * shifts)
    return scales / (variation * variation)


def _estimate_step_scale(tfm, corners, Ai, step, variation=0.01):
    biggest = float(np.abs(step).max())
    if biggest <= np.finfo(np.float64).eps:
        return 0.0
    factor = variation / biggest
    return _max_shift(tfm, corners, Ai, step * factor) / factor


def _golden_section_itk(value_at, a, b, c, state, metric_b=None, epsilon=0.01, max_iterations=20):
    """GradientDescentLineSearchOptimizerv4::GoldenSectionSearch, the recursion as ITK writes it (one probe at a time)."""
    if state["iterations"] > max_iterations:
        return (c + a) / 2.0
    state["iterations"] += 1
    resphi = 2.0 - (1.0 + np.sqrt(5.0)) / 2.0
    x = b + resphi * (c - b) if (c - b) > (b - a) else b - resphi * (b - a)
    if abs(c - a) < epsilon * (abs(b) + abs(x)):
        return (c + a) / 2.0
    if metric_b is None:
        metric_b = value_at(b)
    metric_x = value_at(x)
    if metric_x < metric_b:
        if (c - b) > (b - a):
            return _golden_section_itk(value_at, b, x, c, state, metric_x, epsilon, max_iterations)
        return _golden_section_itk(value_at, a, x, b, state, metric_x, epsilon, max_iterations)
    if (c - b) > (b - a):
        return _golden_section_itk(value_at, a, b, x, state, metric_b, epsilon, max_iterations)
    return _golden_section_itk(value_at, x, b, c, state, metric_b, epsilon, max_iterations)


def registration(fixed, moving, reg_method="similarity", optimiser="gradient_descent", shrink_factors=(8, 2, 1), smooth_sigmas=(4, 2, 0),
                 sampling_rate=0.25, number_of_iterations=50, seed=42, itk_sampling=True, return_best=False, metric="mean_squares"):
    """fixed, moving: oracle.Vol (float32 [Z, Y, X] + geometry).  -> dict(parameters, levels=[dict(values, parameters per
    iteration, learning_rates, scales, stop)], init_matrix, init_offset, evaluations).  mean_squares only (the pipelines' metric).
    itk_sampling=False: samples on the lattice and the interpolant's gradient (the product's opt-out), for A/B tests."""
    from oracle import oracle as O

    kind = {"translation": "translation", "rigid": "rigid", "similarity": "similarity", "affine": "affine"}[reg_method.lower()]
    fD = np.asarray(fixed.direction, dtype=np.float64).reshape(3, 3)
    mD = np.asarray(moving.direction, dtype=np.float64).reshape(3, 3)

    def centre(vol, D):
        n = np.asarray(vol.size, dtype=np.float64)
        return np.asarray(vol.origin) + D @ (np.asarray(vol.spacing) * (n - 1.0) / 2.0)

    # CenteredTransformInitializer(fixed, moving, Euler3DTransform(), GEOMETRY): identity rotation, translation between the centres
    Ai, oi = np.eye(3), centre(moving, mD) - centre(fixed, fD)
    tfm = OracleTransform(kind)
    generator = MersenneTwister(seed) if itk_sampling else None
    levels, max_step, evaluations = [], None, 0
    for shrink, sigma in zip(shrink_factors, smooth_sigmas):
        f_l = O.discrete_gaussian(fixed, sigma * sigma) if sigma > 0 else fixed
        m_l = O.discrete_gaussian(moving, sigma * sigma) if sigma > 0 else moving
        vsize, vspacing, vorigin, vdir = _shrunk_virtual_domain(fixed.size, fixed.spacing, fixed.origin, fixed.direction, shrink)
        stride = int(np.ceil(1.0 / sampling_rate)) if sampling_rate < 1.0 else 1
        nv = int(vsize[0] * vsize[1] * vsize[2])
        lin = np.arange(0, nv, stride, dtype=np.int64)
        idx = np.stack([lin % vsize[0], (lin // vsize[0]) % vsize[1], lin // (vsize[0] * vsize[1])], axis=1).astype(np.float64)
        points = vorigin[None, :] + (idx * vspacing[None, :]) @ vdir.T
        if generator is not None:       # SetMetricSamplePoints: each physical coordinate + N(0, 1) * spacing / 3, raster order
            points = points + generator.normal_variates(3 * len(points)).reshape(len(points), 3) * (vspacing / 3.0)[None, :]
        gimg = O.gradient_recursive_gaussian(m_l).astype(np.float64) if itk_sampling else None
        level_metric = _PhysicalMeanSquares(f_l, m_l, points, Ai, oi, gimg, metric)
        corners = _corner_points(vsize, vspacing, vorigin, vdir)
        # ---- StartOptimization ----
        scales = _estimate_scales(tfm, corners, Ai)
        if max_step is None:            # "if the user hasn't set this, assign the default" -- once per optimiser object
            max_step = float(vspacing.min())
        energies, rec = [], {"values": [], "parameters": [], "learning_rates": [], "scales": scales.tolist(), "stop": "iterations"}
        learning_rate, best_value, best_p = 1.0, np.inf, tfm.p.copy()
        for it in range(number_of_iterations):
            value, derivative = level_metric.value_and_derivative(tfm)
            if not np.isfinite(value):
                if it == 0:
                    raise RuntimeError("no valid sample points")
                rec["stop"] = "no overlap"
                tfm.p = previous
                break
            rec["values"].append(value)
            rec["parameters"].append(tfm.p.copy())
            if value < best_value:
                best_value, best_p = value, tfm.p.copy()
            energies.append(value)
            if _window_convergence_itk(energies) <= 1e-6:
                rec["stop"] = "converged"
                break
            gradient = derivative / scales                                  # ModifyGradientByScales
            if it == 0:                                                     # EstimateLearningRate (Once)
                step_scale = _estimate_step_scale(tfm, corners, Ai, gradient)
                learning_rate = max_step / step_scale if step_scale > np.finfo(np.float64).eps else 1.0
            if optimiser == "gradient_descent_line_search":
                def value_at(rate):
                    return level_metric.value(tfm, tfm.updated(rate * gradient))

                learning_rate = _golden_section_itk(value_at, 0.0 * learning_rate, learning_rate, 5.0 * learning_rate, {"iterations": 0})
            rec["learning_rates"].append(learning_rate)
            previous = tfm.p.copy()
            tfm.p = tfm.updated(learning_rate * gradient)                  # UpdateTransformParameters(m_Gradient)
        if return_best and level_metric.value(tfm) > best_value:
            tfm.p = best_p
        rec["final"] = tfm.p.copy()
        evaluations += level_metric.evaluations
        levels.append(rec)
    return {"parameters": tfm.p.copy(), "levels": levels, "init_matrix": Ai, "init_offset": oi, "evaluations": evaluations,
            "matrix_offset": _total_matrix_offset(tfm, Ai, oi)}


def _total_matrix_offset(tfm, Ai, oi):
    """CompositeTransform([initial, optimised]) as q = A p + off."""
    M, t = tfm.matrix_translation()
    return Ai @ M, Ai @ (t + tfm.c - M @ tfm.c) + oi
