"""Drop-in for platipy/imaging/registration/linear.py:50-260 (linear_registration).

What the reference delegates to ITKv4's ImageRegistrationMethod is rebuilt here around ONE GPU
kernel: the mean-squares metric and its gradient over the regularly sampled virtual domain
(pp_meansq_affine_f32, a 14-number fp64 reduction); everything else -- transform models,
physical-shift parameter scales, learning-rate estimation, gradient descent with ITK's window
convergence test, optional golden-section line search, L-BFGS-B -- is host arithmetic on <= 12
parameters, as it is in ITK.

ITK semantics kept (round 6: they are the DEFAULTS; each was an opt-in or a deviation before):
  * REGULAR sampling takes every ceil(1/rate)-th voxel of the shrunk fixed grid and moves each sample point by ITK's seeded
    sub-voxel jitter (ItkRegularJitter below: the Mersenne-Twister stream of SetMetricSamplingPercentage(rate, seed=42),
    pp_linear_set_sample_jitter); `itk_sampling=False` turns the jitter off (samples on the lattice);
  * the moving-image gradient is ITK's filtered gradient image (GradientRecursiveGaussianImageFilter, sigma = the moving
    image's largest spacing, linearly interpolated at the mapped point: ImageToImageMetricv4's default); with
    `itk_sampling=False` it is the analytic gradient of the trilinear interpolant;
  * every step, line-search probe and scale estimate goes through the transform's UpdateTransformParameters: the versor family
    (rigid, similarity, scale-versor, scale-skew-versor) COMPOSES an axis-angle rotation onto its versor, all other
    parameters are added (transform._versor_update);
  * each level ends at its LAST point (SimpleITK leaves returnBestParametersAndValue off); `return_best_parameters=True` is
    SetOptimizerAs...(returnBestParametersAndValue=True);
  * the learning rate is estimated once per level from m_MaximumStepSizeInPhysicalUnits, which ITK assigns ONCE per optimiser
    -- the smallest virtual spacing of the FIRST level -- and the convergence value is ITK's two-control-point B-spline
    approximation of the energy window, not a least-squares slope (_window_convergence).
Remaining deliberate deviations:
  * a step that leaves the overlap ends the level at the previous point (ITK's metric would warn and return its maximum);
  * all four metrics of the reference run on the GPU: "mean_squares", "correlation", "mattes_mi" (50 bins, fixed
    zero-order / moving cubic-B-spline Parzen windows as itk::MattesMutualInformationImageToImageMetricv4; joint
    histogram in 64-bit fixed point, so results do not depend on scheduling) and "joint_hist_mi" (20 bins, joint PDF
    smoothed with ITK's Gaussian operator of variance 1.5; the derivative differences the smoothed log-PDFs between
    neighbouring moving-bin centres -- this build's estimator under ITK's parameters, not a restatement of ITK's
    interpolated-PDF derivative).  Intensity ranges are taken inside the masks when masks are given, as ITK does;
  * the "exhaustive" optimiser walks the grid of SetOptimizerAsExhaustive (2 n_i + 1 steps per parameter, step length in
    units of the physical-shift scales) in batches of 16 evaluations per launch; numberOfSteps defaults to the reference's
    six tens and can be passed (exhaustive_steps); grids above exhaustive_max_evaluations (EXHAUSTIVE_MAX_EVALUATIONS) raise
    instead of running for days (the reference itself says "use is not currently recommended").
"""
import os
import threading

import numpy as np
import torch

from .. import runtime
from ..image import as_image, cast_tensor
from ..transform import (
    AffineTransform,
    CompositeTransform,
    Euler3DTransform,
    FullAffineTransform,
    ScaleSkewVersor3DTransform,
    ScaleTransform,
    ScaleVersor3DTransform,
    Similarity3DTransform,
    TranslationTransform,
    VersorRigid3DTransform,
    _Parametrised,
)
from .utils import apply_transform, discrete_gaussian

_MODELS = {
    "translation": TranslationTransform,
    "similarity": Similarity3DTransform,
    "affine": FullAffineTransform,
    "rigid": VersorRigid3DTransform,
    "scale": ScaleTransform,
    "scaleversor": ScaleVersor3DTransform,
    "scaleskewversor": ScaleSkewVersor3DTransform,
}
_METRICS = ("mean_squares", "correlation", "mattes_mi", "joint_hist_mi")
MI_BINS = {"mattes_mi": 50, "joint_hist_mi": 20}        # SimpleITK's defaults (numberOfHistogramBins)
JOINT_PDF_SMOOTHING_VARIANCE = 1.5                        # SetMetricAsJointHistogramMutualInformation default
EXHAUSTIVE_MAX_EVALUATIONS = 4_000_000


def _i2p(image):
    d = np.asarray(image.direction, dtype=np.float64).reshape(3, 3)
    return d * np.asarray(image.spacing, dtype=np.float64)[None, :]


def _p2i(image):
    return np.linalg.inv(_i2p(image))


def centered_transform_initializer(fixed_image, moving_image):
    """sitk.CenteredTransformInitializer(fixed, moving, Euler3DTransform(), GEOMETRY) (linear.py:129-131):
    centre of rotation = geometric centre of the fixed image, translation = moving centre - fixed centre."""
    def centre(img):
        n = np.asarray(img.GetSize(), dtype=np.float64)
        return np.asarray(img.origin) + _i2p(img) @ ((n - 1.0) / 2.0)

    t = Euler3DTransform(center=centre(fixed_image))
    p = np.zeros(6)
    p[3:6] = centre(moving_image) - centre(fixed_image)
    t.SetParameters(p)
    return t


def _shrink_geometry(image, factor):
    """itk::ShrinkImageFilter::GenerateOutputInformation: size floor(n/f), spacing*f, same physical centre."""
    n = np.asarray(image.GetSize(), dtype=np.float64)
    f = np.broadcast_to(np.asarray(factor, dtype=np.float64), (3,))
    size = np.maximum(1, np.floor(n / f)).astype(int)
    spacing = np.asarray(image.spacing) * f
    d = np.asarray(image.direction, dtype=np.float64).reshape(3, 3)
    centre_in = np.asarray(image.origin) + _i2p(image) @ ((n - 1.0) / 2.0)
    centre_out = np.asarray(image.origin) + (d * spacing[None, :]) @ ((size - 1.0) / 2.0)
    origin = np.asarray(image.origin) + (centre_in - centre_out)
    return size, spacing, origin, d


class _MeanSquares:
    """value / gradient of the similarity metric for one pyramid level: "mean_squares"
    (itk::MeanSquaresImageToImageMetricv4) or "correlation" (itk::CorrelationImageToImageMetricv4,
    -corr^2), both from one GPU reduction over the sampled virtual domain."""

    def __init__(self, ctx, fixed, moving, vsize, vspacing, vorigin, vdir, initial, sampling_rate, fixed_mask, moving_mask,
                 metric="mean_squares"):
        self.metric = metric
        self.ctx = ctx
        self.fixed, self.moving = fixed, moving
        self.ft = fixed.tensor.contiguous()
        self.mt = moving.tensor.contiguous()
        self.vsize = [int(v) for v in vsize]
        self.i2p_v = vdir * vspacing[None, :]
        self.o_v = vorigin
        self.stride = int(np.ceil(1.0 / sampling_rate)) if sampling_rate < 1.0 else 1   # REGULAR: every ceil(1/p)-th voxel
        self.initial = initial
        self.Ai, self.oi = initial.matrix_offset()
        p2i_f = _p2i(fixed)
        self.Af = p2i_f @ self.i2p_v
        self.bf = p2i_f @ (self.o_v - np.asarray(fixed.origin))
        self.p2i_m = _p2i(moving)
        self.o_m = np.asarray(moving.origin)
        self.fmask = None if fixed_mask is None else fixed_mask.tensor.to(torch.uint8).contiguous()
        self.mmask = None if moving_mask is None else moving_mask.tensor.to(torch.uint8).contiguous()
        n = np.asarray(self.vsize, dtype=np.float64) - 1.0
        corners_idx = np.array([[i, j, k] for k in (0.0, n[2]) for j in (0.0, n[1]) for i in (0.0, n[0])])
        self.corners = self.o_v[None, :] + corners_idx @ self.i2p_v.T     # 8 physical corner points of the virtual domain
        self.min_spacing = float(np.min(vspacing))
        self.evaluations = 0
        self.bins = None
        if metric in MI_BINS:
            from .._lib import MI_JOINT, MI_MATTES, MiBins

            nb, pad = MI_BINS[metric], 2
            # intensity range of each image INSIDE its mask when one is given (both ITK v4 MI metrics' Initialize() walk
            # the image and skip points outside the mask); an empty mask leaves the whole image
            f_lo, f_hi = self._intensity_range(ctx, self.ft, self.fmask)
            m_lo, m_hi = self._intensity_range(ctx, self.mt, self.mmask)
            b = MiBins()
            b.nbins, b.kernel = nb, (MI_MATTES if metric == "mattes_mi" else MI_JOINT)
            # itk::MattesMutualInformation...::Initialize: bin = (max - min) / (bins - 2 padding), normalised min = min / bin - padding
            b.f_bin = max((f_hi - f_lo) / (nb - 2 * pad), 1e-30)
            b.m_bin = max((m_hi - m_lo) / (nb - 2 * pad), 1e-30)
            b.f_norm_min = f_lo / b.f_bin - pad
            b.m_norm_min = m_lo / b.m_bin - pad
            self.bins = b

    @staticmethod
    def _intensity_range(ctx, image, mask):
        if mask is not None:
            inside = mask != 0
            if bool(inside.any()):
                inf = torch.tensor(float("inf"), dtype=image.dtype, device=image.device)
                return float(torch.where(inside, image, inf).min()), float(torch.where(inside, image, -inf).max())
        return ctx.minmax(image, image.numel())

    def total(self, model, params):
        """(A, off) of initial o model(params): q = A p + off."""
        A, t = model.decode(params)
        off = t + model.center - A @ model.center
        return self.Ai @ A, self.Ai @ off + self.oi

    def index_map(self, model, params):
        A, off = self.total(model, params)
        Am = self.p2i_m @ A @ self.i2p_v
        bm = self.p2i_m @ (A @ self.o_v + off - self.o_m)
        return Am, bm

    def raw(self, model, params):
        """-> (value, d value / d (Am row-major, bm)) in index space."""
        Am, bm = self.index_map(model, params)
        self.evaluations += 1
        args = (self.ft, self.fixed.GetSize(), self.mt, self.moving.GetSize(), self.Af.ravel(), self.bf, Am.ravel(), bm, self.vsize,
                self.stride, self.fmask, self.mmask)
        if self.bins is not None:
            hist, count = self.ctx.mi_histogram(*args[:10], self.bins, self.fmask, self.mmask)
            if count <= 0:
                raise RuntimeError("linear_registration: no valid sample points (images do not overlap)")
            value, table = self._mi_value_and_table(hist, count)
            g = self.ctx.mi_gradient(*args[:10], self.bins, table, self.fmask, self.mmask)
            return value, np.asarray(g)
        if self.metric == "mean_squares":
            r = self.ctx.meansq_affine(*args)
            if r[1] <= 0:
                raise RuntimeError("linear_registration: no valid sample points (images do not overlap)")
            return r[0] / r[1], np.asarray(r[2:14]) / r[1]
        r = np.asarray(self.ctx.corr_moments_affine(*args))
        n = r[0]
        if n <= 0:
            raise RuntimeError("linear_registration: no valid sample points (images do not overlap)")
        fbar, mbar = r[1] / n, r[2] / n
        sff, smm, sfm = r[3] - n * fbar * fbar, r[4] - n * mbar * mbar, r[5] - n * fbar * mbar
        if sff <= 1e-300 or smm <= 1e-300:
            return 0.0, np.zeros(12)
        G, FG, MG = r[6:18], r[18:30], r[30:42]
        dsfm = FG - fbar * G                       # sum (f - fbar) dm
        dsmm = 2.0 * (MG - mbar * G)               # d sum (m - mbar)^2
        value = -(sfm * sfm) / (sff * smm)
        grad = -(2.0 * sfm / (sff * smm) * dsfm - (sfm * sfm) / (sff * smm * smm) * dsmm)
        return value, grad

    def _mi_value_and_table(self, hist, count):
        """Joint histogram -> (negative mutual information, per-bin score table for the gradient pass)."""
        eps = 1e-16
        if self.metric == "mattes_mi":
            P = hist / hist.sum()
        else:
            from scipy.ndimage import correlate1d

            from .._lib import gauss_taps

            taps = np.asarray(gauss_taps(JOINT_PDF_SMOOTHING_VARIANCE, 0.01, 32, lib=self.ctx.lib))   # ITK's DiscreteGaussian on the PDF image
            P = correlate1d(correlate1d(hist / count, taps, axis=0, mode="nearest"), taps, axis=1, mode="nearest")
        PF, PM = P.sum(1), P.sum(0)
        ok = (P > eps) & (PF[:, None] > eps) & (PM[None, :] > eps)
        safe = np.where(ok, P, 1.0)
        value = -float((np.where(ok, P * np.log(safe / np.where(ok, PF[:, None] * PM[None, :], 1.0)), 0.0)).sum())
        ratio = np.where(ok, np.log(safe / np.where(ok, np.broadcast_to(PM[None, :], P.shape), 1.0)), 0.0)
        # Mattes: d value / d mu = sum_s sum_k B3'(k - u_s) table[f_s][k] (grad M . dT/dmu), table = log(P / PM) / (N bin)
        # joint:  d value / d mu = -(1/N) sum_s d/dm [log P - log PM](f_s, m_s) (grad M . dT/dmu), differenced between bin centres
        scale = 1.0 / (count * self.bins.m_bin)
        table = ratio * scale if self.metric == "mattes_mi" else -ratio * scale
        return value, table

    def value(self, model, params):
        return self.raw(model, params)[0] if self.bins is None else self.values(model, [params])[0]

    def values(self, model, params_list):
        """Metric values only for up to 16 parameter vectors in one launch (line search); a candidate without any
        valid sample point is +inf."""
        maps = [self.index_map(model, p) for p in params_list]
        self.evaluations += len(maps)
        if self.bins is not None:        # one histogram pass per candidate; the value is host arithmetic on <= 64 x 64 numbers
            out = []
            for Am, bm in maps:
                hist, count = self.ctx.mi_histogram(self.ft, self.fixed.GetSize(), self.mt, self.moving.GetSize(), self.Af.ravel(), self.bf,
                                                    Am.ravel(), bm, self.vsize, self.stride, self.bins, self.fmask, self.mmask)
                out.append(self._mi_value_and_table(hist, count)[0] if count > 0 else float("inf"))
            return out
        r = self.ctx.metric_values_affine(0 if self.metric == "mean_squares" else 1, self.ft, self.fixed.GetSize(), self.mt,
                                          self.moving.GetSize(), self.Af.ravel(), self.bf, [m[0] for m in maps], [m[1] for m in maps],
                                          self.vsize, self.stride, self.fmask, self.mmask)
        out = []
        for row in np.asarray(r):
            if self.metric == "mean_squares":
                out.append(row[0] / row[1] if row[1] > 0 else float("inf"))
                continue
            n = row[0]
            if n <= 0:
                out.append(float("inf"))
                continue
            fbar, mbar = row[1] / n, row[2] / n
            sff, smm, sfm = row[3] - n * fbar * fbar, row[4] - n * mbar * mbar, row[5] - n * fbar * mbar
            out.append(0.0 if sff <= 1e-300 or smm <= 1e-300 else -(sfm * sfm) / (sff * smm))
        return out

    def value_and_gradient(self, model, params):
        value, g_idx = self.raw(model, params)
        params = np.asarray(params, dtype=np.float64)
        grad = np.zeros(len(params))
        for i in range(len(params)):                     # chain rule through params -> (Am, bm), numerically
            h = 1e-6 * max(1.0, abs(params[i]))
            pp, pm = params.copy(), params.copy()
            pp[i] += h
            pm[i] -= h
            Ap, bp = self.index_map(model, pp)
            An, bn = self.index_map(model, pm)
            d = np.concatenate([(Ap - An).ravel(), bp - bn]) / (2 * h)
            grad[i] = float(g_idx @ d)
        return value, grad

    # -- itk::RegistrationParameterScalesFromPhysicalShift over the 8 corners ------------
    def max_shift(self, model, params, delta):
        A0, o0 = self.total(model, params)
        A1, o1 = self.total(model, model.update(params, delta))     # ScalesFromShiftBase::ComputeSampleShifts -> UpdateTransformParameters
        d = (self.corners @ (A1 - A0).T) + (o1 - o0)[None, :]
        return float(np.sqrt((d ** 2).sum(1)).max())

    def scales(self, model, params, variation=0.01):
        n = len(params)
        s = np.zeros(n)
        for i in range(n):
            dlt = np.zeros(n)
            dlt[i] = variation
            s[i] = self.max_shift(model, params, dlt)
        nz = s[s > 1e-12]
        fill = nz.min() if nz.size else 1.0
        s[s <= 1e-12] = fill
        return (s * s) / (variation * variation)

    def step_scale(self, model, params, step, variation=0.01):
        m = float(np.max(np.abs(step)))
        if m <= 1e-300:
            return 0.0
        factor = variation / m
        return self.max_shift(model, params, step * factor) / factor


def _window_convergence(values, window):
    """itk::Function::WindowConvergenceMonitoringFunction::GetConvergenceValue: the last `window` energies, divided by the
    sum of their magnitudes, at t = i / (window - 1), approximated by BSplineScatteredDataPointSetToImageFilter (spline order
    1, two control points, one level).  That filter is the Lee / Wolberg / Shin scattered-data update, not a least-squares
    fit: with the hat weights w0 = 1 - t, w1 = t of a sample, control point k = sum(w_k^3 e / (w0^2 + w1^2)) / sum(w_k^2);
    the convergence value is minus the slope between the two control points.  A sample on the end of the parametric domain
    is moved inside by the filter's epsilon (1e-3 of its 0.1 grid spacing).  +inf until the window is full."""
    if len(values) < window:
        return float("inf")
    e = np.asarray(values[-window:], dtype=np.float64)
    tot = np.abs(e).sum()
    if tot == 0.0:
        return 0.0
    e = e / tot
    t = np.arange(window, dtype=np.float64) / (window - 1)
    t = np.where(np.abs(t - 1.0) <= 1e-4, 1.0 - 1e-4, t)
    w = np.stack([1.0 - t, t])
    lattice = (e[None, :] * w ** 3 / (w ** 2).sum(0)[None, :]).sum(1) / (w ** 2).sum(1)
    return -float(lattice[1] - lattice[0])


# The gradient-descent optimisers run inside the library (pp_linear.hip: one call per level, no interpreter between
# the ~1000 launches) for the built-in transform models; the Python loop below is the same algorithm and serves
# user-defined _Parametrised models and scipy's L-BFGS-B.
NATIVE_OPTIMISER = True
_NATIVE_MODEL = {TranslationTransform: 0, VersorRigid3DTransform: 1, Similarity3DTransform: 2, ScaleTransform: 3,
                 FullAffineTransform: 4, Euler3DTransform: 5, ScaleVersor3DTransform: 6, ScaleSkewVersor3DTransform: 7}


def _optimise_level_native(ctx, ms, model, params, opt, number_of_iterations, verbose, max_step, return_best, record):
    """pp_linear_optimize_f32 on the level described by `ms`."""
    from .._lib import ERR_NO_OVERLAP, LINREG_RETURN_BEST, LinregLevel, PlatipyAmdError

    lv = LinregLevel()
    lv.model = _NATIVE_MODEL[type(model)]
    lv.metric = 0 if ms.metric == "mean_squares" else 1
    lv.optimizer = 1 if opt == "gradient_descent_line_search" else 0
    lv.iterations = int(number_of_iterations)
    lv.vsize[:] = [int(v) for v in ms.vsize]
    lv.stride = int(ms.stride)
    lv.speculation = int(os.environ.get("PP_LINE_SEARCH_SPECULATION", LINE_SEARCH_SPECULATION))
    lv.v_i2p[:] = ms.i2p_v.ravel().tolist()
    lv.v_origin[:] = np.asarray(ms.o_v, dtype=np.float64).tolist()
    lv.f_p2i[:] = _p2i(ms.fixed).ravel().tolist()
    lv.f_origin[:] = np.asarray(ms.fixed.origin, dtype=np.float64).tolist()
    lv.m_p2i[:] = ms.p2i_m.ravel().tolist()
    lv.m_origin[:] = ms.o_m.tolist()
    lv.init_matrix[:] = np.asarray(ms.Ai, dtype=np.float64).ravel().tolist()
    lv.init_offset[:] = np.asarray(ms.oi, dtype=np.float64).tolist()
    lv.center[:] = np.asarray(model.center, dtype=np.float64).tolist()
    lv.v_min_spacing = float(max_step)
    lv.flags = LINREG_RETURN_BEST if return_best else 0
    try:
        out, stats, history = ctx.linear_optimize(ms.ft, ms.fixed.GetSize(), ms.mt, ms.moving.GetSize(), lv, params, ms.fmask, ms.mmask,
                                                  history=number_of_iterations if (verbose or record is not None) else 0)
    except PlatipyAmdError as e:
        if getattr(e, "code", 0) == ERR_NO_OVERLAP:
            raise RuntimeError("linear_registration: no valid sample points (images do not overlap)") from e
        raise
    ms.evaluations += stats.evaluations
    if verbose:
        for it, value in enumerate(history):
            print("{0:3} = {1:10.5f}".format(it, value))
    if record is not None:
        record.append({"values": list(history), "iterations": int(stats.iterations), "stop": int(stats.stop),
                       "learning_rate": float(stats.learning_rate), "parameters": [float(v) for v in out]})
    return np.asarray(out, dtype=np.float64)


# How many levels of the golden-section decision tree are evaluated per launch (2^depth - 1 learning rates, + the
# bracket's middle point on the first round: 16 at depth 4 = one pp_metric_values_affine_f32 call).
LINE_SEARCH_SPECULATION = 4   # measured best (profiles/round3_linear_speculation.txt: affine stage 32.4 / 34.0 / 38.0 / 65.3 ms at
#                               depth 4 / 3 / 2 / 1); PP_LINE_SEARCH_SPECULATION overrides it for such sweeps


def _golden_section(fbatch, a, b, c, eps=0.01, max_iter=20, depth=None):
    """itk::GradientDescentLineSearchOptimizerv4::GoldenSectionSearch on the learning rate (a < b < c).

    The search is sequential -- each probe depends on the previous comparison -- and every probe is a GPU launch
    plus a read-back whose cost is all latency.  So the next `depth` levels of its decision tree are probed
    speculatively in ONE batched launch (`fbatch(list of learning rates) -> list of values`), then the search
    walks the tree with the values in hand.  The probes taken, their order and the result are exactly those of
    the sequential search (depth = 1)."""
    depth = LINE_SEARCH_SPECULATION if depth is None else depth
    resphi = 2.0 - (1.0 + np.sqrt(5.0)) / 2.0
    known = {}

    def probe(a, b, c):
        return b + resphi * (c - b) if (c - b) > (b - a) else b - resphi * (b - a)

    def children(a, b, c, x):
        """-> (state if f(x) < f(b), state otherwise)"""
        if (c - b) > (b - a):
            return (b, x, c), (a, b, x)
        return (a, x, b), (x, b, c)

    def speculate(a, b, c, levels, left, want):
        if levels == 0 or left == 0:
            return
        x = probe(a, b, c)
        if abs(c - a) < eps * (abs(b) + abs(x)):
            return
        if x not in known and x not in want:
            want.append(x)
        lo, hi = children(a, b, c, x)
        speculate(*lo, levels - 1, left - 1, want)
        speculate(*hi, levels - 1, left - 1, want)

    fb = None
    for it in range(max_iter):
        x = probe(a, b, c)
        if abs(c - a) < eps * (abs(b) + abs(x)):
            return (c + a) / 2.0
        if x not in known or (fb is None and b not in known):
            want = []
            speculate(a, b, c, depth, max_iter - it, want)
            if fb is None and b not in known and b not in want:
                want.append(b)
            for key, val in zip(want, fbatch(want)):
                known[key] = val
        fx = known[x]
        if fb is None:
            fb = known[b]
        lo, hi = children(a, b, c, x)
        if fx < fb:
            (a, b, c), fb = lo, fx
        else:
            a, b, c = hi
    return (c + a) / 2.0


def _exhaustive(ms, model, params, number_of_steps, step_length, verbose, max_evaluations=None):
    """itk::ExhaustiveOptimizerv4 as SetOptimizerAsExhaustive(numberOfSteps, stepLength=1.0) drives it (reference
    linear.py:215-222): every point of the grid initial + (k_i - n_i) * stepLength * scale_i, k_i = 0 .. 2 n_i, with the
    optimiser scales from physical shift; the best point wins.  Evaluated 16 grid points per launch."""
    n = len(params)
    if len(number_of_steps) != n:
        raise ValueError(f"exhaustive: numberOfSteps has {len(number_of_steps)} entries, the transform has {n} parameters "
                         "(ITK raises here too; the reference passes six)")
    total = int(np.prod([2 * k + 1 for k in number_of_steps], dtype=np.float64))
    limit = EXHAUSTIVE_MAX_EVALUATIONS if max_evaluations is None else int(max_evaluations)
    if total > limit:
        raise ValueError(f"exhaustive: the grid has {total:,} points (> EXHAUSTIVE_MAX_EVALUATIONS = {limit:,}); the reference would "
                         "evaluate them one by one -- pass fewer exhaustive_steps or a larger exhaustive_max_evaluations")
    scales = ms.scales(model, params)
    base = np.asarray(params, dtype=np.float64)
    best_v, best_p = float("inf"), base.copy()
    axes = [np.arange(-k, k + 1) for k in number_of_steps]
    batch = []

    def flush():
        nonlocal best_v, best_p
        if not batch:
            return
        for p, v in zip(batch, ms.values(model, batch)):
            if v < best_v:
                best_v, best_p = v, p
        batch.clear()

    import itertools

    for idx in itertools.product(*axes):              # last parameter fastest, like ITK's odometer
        batch.append(base + np.asarray(idx, dtype=np.float64) * step_length * scales)
        if len(batch) == 16:
            flush()
    flush()
    if verbose:
        print(f"exhaustive: {total} evaluations, best value {best_v:.6f}")
    return best_p


def linear_registration(
    fixed_image,
    moving_image,
    fixed_structure=None,
    moving_structure=None,
    reg_method="similarity",
    metric="mean_squares",
    optimiser="gradient_descent",
    shrink_factors=[8, 2, 1],
    smooth_sigmas=[4, 2, 0],
    sampling_rate=0.25,
    final_interp=2,
    number_of_iterations=50,
    default_value=None,
    verbose=False,
    exhaustive_steps=None,
    exhaustive_step_length=1.0,
    exhaustive_max_evaluations=None,
    itk_sampling=True,
    sampling_seed=42,
    return_best_parameters=False,
):
    """Initial linear registration between two images (reference registration/linear.py:50-260).

    Returns (registered_image, CompositeTransform([initial_centering_transform, optimised_transform])).

    `itk_sampling` (default True = the reference's semantics; False is the opt-out that samples on the lattice): (i) the REGULAR
    sample points carry ITK's seeded sub-voxel jitter -- registration.SetMetricSamplingPercentage(sampling_rate, seed=42),
    linear.py:151 -- drawn from one Mersenne-Twister stream over the levels (`sampling_seed`, the reference's 42); (ii) the
    moving-image gradient is the linear interpolation of ITK's filtered gradient image (GradientRecursiveGaussianImageFilter
    with sigma = the moving image's largest spacing, NormalizeAcrossScale, once per level: ImageToImageMetricv4's default)
    instead of the derivative of the intensity interpolant.  For every metric and optimiser.  `return_best_parameters` is
    SetOptimizerAsGradientDescent[LineSearch](..., returnBestParametersAndValue): SimpleITK's default False returns each
    level's last point.  After the call `linear_registration.last_levels` holds, per level, the optimiser's record
    (metric value per iteration, iterations taken, stop reason, learning rate, parameters) of the gradient-descent optimisers.

    The last three arguments are extensions for optimiser="exhaustive" (the reference hard-codes numberOfSteps = [10] * 6 at
    linear.py:221, 21^6 = 85.8 M evaluations on a six-parameter model, and says itself that "use is not currently
    recommended"): `exhaustive_steps` = SetOptimizerAsExhaustive's numberOfSteps, one entry per transform parameter
    (default: the reference's six tens), `exhaustive_step_length` its stepLength, `exhaustive_max_evaluations` the largest
    grid this call may walk (default EXHAUSTIVE_MAX_EVALUATIONS; the reference's own grid needs 85_766_121).
    """
    fixed_image, moving_image = as_image(fixed_image), as_image(moving_image)
    moving_image_type = moving_image.tensor.dtype
    fixed_image = fixed_image.astype(torch.float32)
    moving_image = moving_image.astype(torch.float32)
    ctx = runtime.context(fixed_image.device)

    metric = metric.lower()
    if metric not in _METRICS:
        raise ValueError(f"unknown metric {metric!r}: choose from {_METRICS}")
    initial_transform = centered_transform_initializer(fixed_image, moving_image)

    if isinstance(reg_method, str):
        key = reg_method.lower()
        if key not in _MODELS:
            raise ValueError(
                "You have selected a registration method that does not exist.\n Please select from"
                " Translation, Similarity, Affine, Rigid, ScaleVersor, ScaleSkewVersor")
        model = _MODELS[key]()
    elif isinstance(reg_method, _Parametrised):
        model = reg_method
    elif isinstance(reg_method, AffineTransform):
        model = FullAffineTransform(center=reg_method.center)
        model.SetParameters(np.concatenate([reg_method.matrix.ravel(), reg_method.translation]))
    else:
        raise ValueError("'reg_method' must be either a string (see docs for acceptable registration names), "
                         "or a transform instance.")
    opt = optimiser.lower()
    if opt not in ("gradient_descent", "gradient_descent_line_search", "lbfgsb", "exhaustive"):
        raise ValueError(f"unknown optimiser {optimiser!r}")

    fixed_mask = as_image(fixed_structure) if fixed_structure is not None else None
    moving_mask = as_image(moving_structure) if moving_structure is not None else None
    params = np.asarray(model.GetParameters(), dtype=np.float64)
    record = []
    if not itk_sampling:
        params = _optimise_levels(ctx, None, **_level_args(locals()))
    else:
        jitter_source = _JitterSource(sampling_seed, fixed_image.device)
        try:
            params = _optimise_levels(ctx, jitter_source, **_level_args(locals()))
        finally:
            ctx.set_sample_jitter(None)
            ctx.set_moving_gradient(None)
            jitter_source.close()
    linear_registration.last_levels = record

    model.SetParameters(params)
    output_transform = model
    combined_transform = CompositeTransform([initial_transform, output_transform])     # linear.py:240

    if default_value is None:
        default_value = 0
        if float(moving_image.tensor.min()) <= -1000:
            default_value = -1000
    registered_image = apply_transform(input_image=moving_image, reference_image=fixed_image, transform=combined_transform,
                                       default_value=default_value, interpolator=final_interp)
    registered_image = registered_image.like(cast_tensor(registered_image.tensor, moving_image_type))
    return registered_image, combined_transform


class ItkRegularJitter:
    """The sub-voxel jitter of itk::ImageRegistrationMethodv4's REGULAR sampling (SetMetricSamplePoints): every sample point's
    PHYSICAL coordinate d is moved by GetNormalVariate() * virtual spacing[d] / 3, the variates coming from ONE
    itk::Statistics::MersenneTwisterRandomVariateGenerator seeded by SetMetricSamplingPercentage(rate, seed) -- level after
    level from the same stream, three per sample in raster order.  MT19937 with init_genrand seeding is numpy's legacy
    RandomState; ITK's normal variate is Box-Muller on an open-range and an open-upper-range uniform, the radius' drawn first
    (the test suite holds this against an independent restatement of the published generator and its known first output).
    ITK 5.3 from memory: parity unpinned."""

    def __init__(self, seed=42):
        self._rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)

    def normal_variates(self, count):
        u = self._rs.randint(0, 2 ** 32, size=2 * int(count), dtype=np.uint64).astype(np.float64)
        r = np.sqrt(-2.0 * np.log(1.0 - (u[0::2] + 0.5) * (1.0 / 4294967296.0)))
        return r * np.cos(2.0 * np.pi * (u[1::2] * (1.0 / 4294967296.0)))

    def level(self, vsize, stride, vspacing, vdir):
        """-> [nsamples, 3] float32, virtual-index units (what pp_linear_set_sample_jitter takes)."""
        nv = int(vsize[0]) * int(vsize[1]) * int(vsize[2])
        nsamp = (nv + int(stride) - 1) // int(stride)
        sp = np.asarray(vspacing, dtype=np.float64)
        phys = self.normal_variates(3 * nsamp).reshape(nsamp, 3) * (sp / 3.0)[None, :]
        p2i = np.linalg.inv(np.asarray(vdir, dtype=np.float64).reshape(3, 3) * sp[None, :])
        return np.ascontiguousarray((phys @ p2i.T).astype(np.float32))


PACKED_GRADIENT_MIN_SAMPLES = 200_000

# (device, seed, levels so far) -> device tensor.  Two rules, both learnt the hard way (profiles/round6_gpu_suite.txt: four atlas
# chains started together deviated from the sequential run on 4 of 17 fresh boxes, and again on 1 of 7 with the cache off):
#   * entries are NEVER evicted -- a cached tensor is read by kernels on whatever stream its caller runs on, while the allocator
#     knows only the stream it was allocated on (the rule registration/utils.py::_need_masks follows since ADVICE round 3); past
#     the bounds new levels are simply not cached, runtime.release_all() empties the cache after a device synchronisation;
#   * a level is drawn and uploaded by ONE thread, under the lock: threads that miss the same key together wait for it instead
#     of each uploading a private copy that is freed again when its level ends (the chains that deviated were exactly those
#     running on such private copies; what goes wrong with them was not reconstructed).  A copy that cannot be cached (bounds
#     reached) stays with its _JitterSource until close(), which waits for the stream before the allocator gets it back.
_JITTER_CACHE = {}
_JITTER_CACHE_MAX = int(os.environ.get("PP_JITTER_CACHE_MAX", 64))      # (0: every registration draws and uploads its own levels)
_JITTER_CACHE_MAX_BYTES = 1 << 30
_JITTER_LOCK = threading.Lock()


def release_cached_jitter():
    """Called by runtime.release_all()."""
    with _JITTER_LOCK:
        _JITTER_CACHE.clear()


class _JitterSource:
    """ItkRegularJitter's levels as DEVICE tensors, remembered across registrations: the stream depends only on the seed and on
    the sequence of (virtual size, stride, spacing, direction) of the levels drawn so far, and a multi-atlas run registers every
    atlas onto the same target grid with the same seed -- 1.5 M Mersenne-Twister draws per registration otherwise."""

    def __init__(self, seed, device):
        self.seed, self.device, self.history, self._gen, self._drawn, self._private = int(seed), device, (), None, 0, []

    def _draw(self, vsize, stride, vspacing, vdir):
        if self._gen is None:
            self._gen = ItkRegularJitter(self.seed)
        while self._drawn < len(self.history) - 1:      # levels served from the cache: their variates still have to be consumed
            self._gen.level(*self.history[self._drawn][:3], np.asarray(self.history[self._drawn][3]).reshape(3, 3))
            self._drawn += 1
        host = self._gen.level(vsize, stride, vspacing, vdir)
        t = torch.from_numpy(host).to(self.device)
        if t.is_cuda:
            # The one array of this path that reaches the GPU through a host-to-device copy, and the one the unexplained
            # deviation of concurrently uploaded copies points at (profiles/round6_gpu_suite.txt): hold the upload to the host
            # array's sum -- one reduction per level and geometry -- and repeat it once, loudly, if it does not arrive whole.
            want = float(host.sum(dtype=np.float64))
            tol = 1e-9 * float(np.abs(host).sum(dtype=np.float64)) + 1e-12
            if abs(float(t.sum(dtype=torch.float64)) - want) > tol:
                import warnings

                warnings.warn("platipy_amd: a sample-jitter upload did not arrive whole; uploading it again", RuntimeWarning)
                t = torch.from_numpy(host).to(self.device)
                if abs(float(t.sum(dtype=torch.float64)) - want) > tol:
                    from .._lib import PlatipyAmdError

                    raise PlatipyAmdError("the sample-jitter array could not be uploaded intact")
        self._drawn += 1
        return t

    def level(self, vsize, stride, vspacing, vdir):
        here = (tuple(int(v) for v in vsize), int(stride), tuple(float(v) for v in vspacing), tuple(float(v) for v in np.ravel(vdir)))
        self.history = self.history + (here,)
        key = (str(self.device), self.seed, self.history)
        with _JITTER_LOCK:
            hit = _JITTER_CACHE.get(key)
            if hit is not None:
                return hit
            nv = int(vsize[0]) * int(vsize[1]) * int(vsize[2])
            nbytes = 12 * ((nv + int(stride) - 1) // int(stride))
            held = sum(v.numel() * v.element_size() for v in _JITTER_CACHE.values())
            if len(_JITTER_CACHE) < _JITTER_CACHE_MAX and held + nbytes <= _JITTER_CACHE_MAX_BYTES:
                t = self._draw(vsize, stride, vspacing, vdir)       # (under the lock: the others wait for this upload, then hit)
                _JITTER_CACHE[key] = t
                return t
        t = self._draw(vsize, stride, vspacing, vdir)
        self._private.append(t)
        return t

    def close(self):
        """End of the registration: uncached levels go back to the allocator only once the stream has finished with them."""
        if self._private:
            if torch.device(self.device).type == "cuda":
                torch.cuda.current_stream(torch.device(self.device)).synchronize()
            self._private.clear()


def itk_moving_gradient(ctx, moving):
    """ImageToImageMetricv4's default moving-image gradient source, in moving-INDEX units [3, Z, Y, X] float32:
    itk::GradientRecursiveGaussianImageFilter(sigma = largest spacing, NormalizeAcrossScale, UseImageDirection) -- per component
    d the first-order recursive Gaussian along d, then the zero-order ones along the other axes in increasing order, divided by
    spacing[d]; rotated to physical axes by the direction cosines; then d m / d index = (direction * spacing)^T applied to it.
    With identity direction cosines the division by spacing[d] and the index-unit conversion (times spacing[d]) cancel: the
    filter chain's output IS the per-index gradient, written straight into its plane of the result."""
    src = moving.tensor if moving.tensor.dtype == torch.float32 else moving.tensor.float()
    src = src.contiguous()
    geom = moving.geom()
    sp = np.asarray(moving.spacing, dtype=np.float64)
    sigma = float(sp.max())
    out = torch.empty((3,) + tuple(src.shape), dtype=torch.float32, device=src.device)
    scratch = [torch.empty_like(src), torch.empty_like(src)]
    for d in range(3):
        others = [ax for ax in range(3) if ax != d]
        ctx.recursive_gaussian_pass(src, scratch[0], geom, d, sigma, order=1, normalize_across_scale=True)
        ctx.recursive_gaussian_pass(scratch[0], scratch[1], geom, others[0], sigma, order=0, normalize_across_scale=True)
        ctx.recursive_gaussian_pass(scratch[1], out[d], geom, others[1], sigma, order=0, normalize_across_scale=True)
    D = np.asarray(moving.direction, dtype=np.float64).reshape(3, 3)
    if np.array_equal(D, np.eye(3)):
        return out
    # oblique / flipped grids: physical gradient = D (filtered_d / spacing_d), per-index gradient = (D spacing)^T of that
    to_index = (D * sp[None, :]).T @ D @ np.diag(1.0 / sp)
    mixed = torch.empty_like(out)
    for r in range(3):
        torch.mul(out[0], float(to_index[r, 0]), out=mixed[r])
        mixed[r].add_(out[1], alpha=float(to_index[r, 1])).add_(out[2], alpha=float(to_index[r, 2]))
    return mixed


def _level_args(scope):
    keys = ("fixed_image", "moving_image", "fixed_mask", "moving_mask", "initial_transform", "model", "params", "metric", "opt",
            "shrink_factors", "smooth_sigmas", "sampling_rate", "number_of_iterations", "verbose", "exhaustive_steps",
            "exhaustive_step_length", "exhaustive_max_evaluations", "return_best_parameters", "record")
    return {k: scope[k] for k in keys}


def _optimise_levels(ctx, jitter, fixed_image, moving_image, fixed_mask, moving_mask, initial_transform, model, params, metric, opt,
                     shrink_factors, smooth_sigmas, sampling_rate, number_of_iterations, verbose, exhaustive_steps,
                     exhaustive_step_length, exhaustive_max_evaluations, return_best_parameters=False, record=None):
    """The resolution levels of linear_registration (ImageRegistrationMethodv4's level loop) -> optimised parameters."""
    max_step = None     # GradientDescentOptimizerBasev4::m_MaximumStepSizeInPhysicalUnits: assigned at the first StartOptimization
    gradient_of = {}    # id(level's moving tensor) -> its filtered gradient image (levels without smoothing share the image)
    for level, (shrink, sigma) in enumerate(zip(shrink_factors, smooth_sigmas)):
        # ImageRegistrationMethodv4::InitializeRegistrationAtEachLevel: smooth both (physical sigma), shrink the virtual domain
        f_l = discrete_gaussian(fixed_image, sigma * sigma) if sigma > 0 else fixed_image
        m_l = discrete_gaussian(moving_image, sigma * sigma) if sigma > 0 else moving_image
        vsize, vspacing, vorigin, vdir = _shrink_geometry(fixed_image, shrink)
        ms = _MeanSquares(ctx, f_l, m_l, vsize, vspacing, vorigin, vdir, initial_transform, sampling_rate, fixed_mask, moving_mask,
                          metric=metric)
        if max_step is None:
            max_step = ms.min_spacing
        if jitter is not None:      # this level's perturbed sample points, for every metric kernel until the next level replaces them
            ctx.set_sample_jitter(jitter.level(ms.vsize, ms.stride, vspacing, vdir))
            # ... and this level's filtered gradient image (ITK filters per level; levels that share the moving image -- no
            # smoothing sigma -- share the result, which is the same numbers)
            key = id(m_l.tensor)
            if key not in gradient_of:
                gradient_of.clear()
                grad = itk_moving_gradient(ctx, m_l)
                # the value + gradient kernel of the mean-squares / correlation metrics gathers (gradient, intensity) as ONE
                # 16-byte element per corner (pp_linear_set_moving_gradient_packed); the MI kernels read the planar image
                # -- when some level's lattice is large enough for the layout to pay for the 2 GB copy that builds it (measured at
                # 512 x 512 x 256: the 128 x 128 x 64 lattice's gradient launch 138 -> 60 us; lattices of 64 x 64 x 32 gain less
                # than the ~1 ms copy)
                packed = None
                largest = max(int(np.prod(_shrink_geometry(fixed_image, f)[0])) for f in shrink_factors) / max(1, ms.stride)
                if metric in ("mean_squares", "correlation") and grad.is_cuda and largest >= PACKED_GRADIENT_MIN_SAMPLES:
                    src = m_l.tensor if m_l.tensor.dtype == torch.float32 else m_l.tensor.float()
                    packed = torch.stack((grad[0], grad[1], grad[2], src), dim=-1)      # [Z, Y, X, 4], one copy kernel
                gradient_of[key] = (m_l.tensor, grad, packed)     # (the image is kept: its id stays its own)
            ctx.set_moving_gradient(gradient_of[key][1], packed=gradient_of[key][2])

        if opt == "lbfgsb":
            from scipy.optimize import fmin_l_bfgs_b

            # optimise in physical-shift units (x = theta * sqrt(scale)) so rotations and translations are commensurate
            root = np.sqrt(ms.scales(model, params))
            start_value = ms.value(model, params)

            def fun(x):
                try:
                    v, g = ms.value_and_gradient(model, x / root)
                except RuntimeError:                      # the trial left the overlap
                    return 10.0 * start_value + 1.0, np.zeros_like(x)
                return v, g / root

            x, _, _ = fmin_l_bfgs_b(fun, params * root, m=50, factr=1e7, pgtol=1e-5, maxiter=number_of_iterations, maxfun=1024)
            if ms.value(model, x / root) <= start_value:
                params = x / root
            continue

        if opt == "exhaustive":
            steps = [10, 10, 10, 10, 10, 10] if exhaustive_steps is None else [int(k) for k in exhaustive_steps]     # linear.py:215-222
            params = _exhaustive(ms, model, params, steps, float(exhaustive_step_length), verbose, exhaustive_max_evaluations)
            continue

        if NATIVE_OPTIMISER and type(model) in _NATIVE_MODEL and ms.bins is None:
            params = _optimise_level_native(ctx, ms, model, params, opt, number_of_iterations, verbose, max_step, return_best_parameters,
                                            record)
            continue

        scales = ms.scales(model, params)
        learning_rate = 1.0
        history = []
        best_value, best_params = float("inf"), params.copy()
        previous, stop = params.copy(), 0
        for it in range(number_of_iterations):
            try:
                value, grad = ms.value_and_gradient(model, params)
            except RuntimeError:
                if it == 0:
                    raise
                params, stop = previous, 2                      # stepped off the overlap: the level ends at the last point with samples
                break
            if value < best_value:
                best_value, best_params = value, params.copy()
            history.append(value)
            if verbose:
                print("{0:3} = {1:10.5f}".format(it, value))
            if _window_convergence(history, 10) <= 1e-6:
                stop = 1
                break
            g = grad / scales                                   # ModifyGradientByScales
            if it == 0:                                         # estimateLearningRate = Once (per StartOptimization: per level)
                ss = ms.step_scale(model, params, -g)
                learning_rate = max_step / ss if ss > np.finfo(np.float64).eps else 1.0
            if opt == "gradient_descent_line_search":
                base = params.copy()

                def trial(es):
                    return ms.values(model, [model.update(base, -e * g) for e in es])

                lr = _golden_section(trial, 0.0, learning_rate, 5.0 * learning_rate)
                learning_rate = lr if lr > 0 else learning_rate
            previous = params
            params = model.update(params, -learning_rate * g)   # m_Metric->UpdateTransformParameters(m_Gradient)
        if return_best_parameters:
            try:
                last = ms.value(model, params)
            except RuntimeError:
                last = float("inf")
            if last > best_value:
                params = best_params
        if record is not None:
            record.append({"values": list(history), "iterations": len(history), "stop": stop, "learning_rate": float(learning_rate),
                           "parameters": [float(v) for v in params]})

    return params
