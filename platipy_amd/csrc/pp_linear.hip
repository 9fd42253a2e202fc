// platipy_amd/csrc/pp_linear.hip -- one resolution level of linear_registration's optimisation, host side.
//
// Replaces what registration.Execute(fixed, moving) does inside one level of sitk.ImageRegistrationMethod as
// configured at platipy/imaging/registration/linear.py:129-238 [ITK-upstream, restated]: ImageRegistrationMethodv4
// with REGULAR sampling, RegistrationParameterScalesFromPhysicalShift over the 8 corners of the virtual domain
// (SetOptimizerScalesFromPhysicalShift, linear.py:231), GradientDescentOptimizerv4 /
// GradientDescentLineSearchOptimizerv4 (learning rate estimated once per level, golden-section search on
// [0, 5] x learning rate, epsilon 0.01, <= 20 probes), convergence window 10 / minimum value 1e-6, the transform's own
// UpdateTransformParameters for every step, probe and scale estimate (the versor family composes its rotation), and -- as
// SimpleITK leaves returnBestParametersAndValue off -- the LAST point of the level as its result.
//
// The metric and its gradient are GPU kernels (pp_fusion.hip); everything here is the optimiser's host logic --
// a few hundred flops per iteration -- kept native so that a registration is one library call per level: no
// interpreter between the ~1000 launches of a level, and none of the caller's global locks held while several
// registrations run side by side on their own HIP streams.  The golden-section search probes the next levels of
// its decision tree speculatively in one batched launch (pp_metric_values_affine_f32); probes, order and result
// are those of the sequential search.
#include <cmath>
#include <limits>
#include <vector>

#include "pp_internal.h"
#include "pp_kernels.h"

namespace {

constexpr double INF = std::numeric_limits<double>::infinity();

void mat_mul(const double* A, const double* B, double* C) {
  double t[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) t[r * 3 + c] = A[r * 3 + 0] * B[0 * 3 + c] + A[r * 3 + 1] * B[1 * 3 + c] + A[r * 3 + 2] * B[2 * 3 + c];
  for (int k = 0; k < 9; ++k) C[k] = t[k];
}

void mat_vec(const double* A, const double* x, double* y) {
  double t[3];
  for (int r = 0; r < 3; ++r) t[r] = A[r * 3 + 0] * x[0] + A[r * 3 + 1] * x[1] + A[r * 3 + 2] * x[2];
  for (int k = 0; k < 3; ++k) y[k] = t[k];
}

// itk::VersorRigid3DTransform::UpdateTransformParameters (inherited by Similarity3D / ScaleVersor3D / ScaleSkewVersor3D): the
// first three entries of the update are an axis-angle rotation (angle = their norm) composed onto the current versor on the
// right; the remaining parameters are added by the caller.
void versor_compose(const double* p, const double* u, double* out) {
  const double x = p[0], y = p[1], z = p[2];
  const double w = std::sqrt(std::fmax(0.0, 1.0 - (x * x + y * y + z * z)));
  const double norm = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  double gx = 0.0, gy = 0.0, gz = 0.0, gw = 1.0;
  if (norm > 0.0) {
    const double f = std::sin(norm / 2.0) / norm;
    gx = u[0] * f, gy = u[1] * f, gz = u[2] * f, gw = std::cos(norm / 2.0);
  }
  double nx = w * gx - z * gy + y * gz + x * gw;
  double ny = z * gx + w * gy - x * gz + y * gw;
  double nz = -y * gx + x * gy + w * gz + z * gw;
  const double nw = -x * gx - y * gy - z * gz + w * gw;
  if (nw < 0.0) nx = -nx, ny = -ny, nz = -nz;   // only the right part is kept and w = +sqrt(1 - |v|^2): same rotation
  out[0] = nx, out[1] = ny, out[2] = nz;
}

bool versor_model(int model) {
  return model == PP_MODEL_VERSOR_RIGID || model == PP_MODEL_SIMILARITY || model == PP_MODEL_SCALE_VERSOR || model == PP_MODEL_SCALE_SKEW_VERSOR;
}

// itk::Transform::UpdateTransformParameters(update, 1): out = p (+) update
void update_parameters(int model, int n, const double* p, const double* u, double* out) {
  double v[3];
  const bool versor = versor_model(model);
  if (versor) versor_compose(p, u, v);
  for (int i = 0; i < n; ++i) out[i] = p[i] + u[i];
  if (versor) out[0] = v[0], out[1] = v[1], out[2] = v[2];
}

void versor_matrix(const double* v, double* R) {
  double x = v[0], y = v[1], z = v[2];
  double n2 = x * x + y * y + z * z;
  if (n2 > 1.0) {  // (rounding only: composed versors have unit norm)
    const double s = 1.0 / std::sqrt(n2);
    x *= s;
    y *= s;
    z *= s;
    n2 = 1.0;
  }
  const double w = std::sqrt(std::fmax(0.0, 1.0 - n2));
  R[0] = 1 - 2 * (y * y + z * z);
  R[1] = 2 * (x * y - z * w);
  R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);
  R[4] = 1 - 2 * (x * x + z * z);
  R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);
  R[7] = 2 * (y * z + x * w);
  R[8] = 1 - 2 * (x * x + y * y);
}

int model_params(int model) {
  switch (model) {
    case PP_MODEL_TRANSLATION: return 3;
    case PP_MODEL_VERSOR_RIGID: return 6;
    case PP_MODEL_SIMILARITY: return 7;
    case PP_MODEL_SCALE: return 3;
    case PP_MODEL_AFFINE: return 12;
    case PP_MODEL_EULER: return 6;
    case PP_MODEL_SCALE_VERSOR: return 9;
    case PP_MODEL_SCALE_SKEW_VERSOR: return 15;
    default: return -1;
  }
}

// parameters -> (A, t) of q = A (p - c) + c + t; the parameter layouts of sitk's transforms
void decode(int model, const double* p, double* A, double* t) {
  static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < 9; ++k) A[k] = I[k];
  t[0] = t[1] = t[2] = 0.0;
  switch (model) {
    case PP_MODEL_TRANSLATION:
      for (int k = 0; k < 3; ++k) t[k] = p[k];
      break;
    case PP_MODEL_VERSOR_RIGID:
      versor_matrix(p, A);
      for (int k = 0; k < 3; ++k) t[k] = p[3 + k];
      break;
    case PP_MODEL_SIMILARITY:
      versor_matrix(p, A);
      for (int k = 0; k < 9; ++k) A[k] *= p[6];
      for (int k = 0; k < 3; ++k) t[k] = p[3 + k];
      break;
    case PP_MODEL_SCALE:
      A[0] = p[0];
      A[4] = p[1];
      A[8] = p[2];
      break;
    case PP_MODEL_AFFINE:
      for (int k = 0; k < 9; ++k) A[k] = p[k];
      for (int k = 0; k < 3; ++k) t[k] = p[9 + k];
      break;
    case PP_MODEL_SCALE_VERSOR:        // itk::ScaleVersor3DTransform::ComputeMatrix: rotation matrix, diagonal += scale - 1
    case PP_MODEL_SCALE_SKEW_VERSOR:   // ... and off-diagonal entries += the six skew terms (additive form)
      versor_matrix(p, A);
      A[0] += p[6] - 1.0;
      A[4] += p[7] - 1.0;
      A[8] += p[8] - 1.0;
      if (model == PP_MODEL_SCALE_SKEW_VERSOR) {
        A[1] += p[9];  A[2] += p[10];
        A[3] += p[11]; A[5] += p[12];
        A[6] += p[13]; A[7] += p[14];
      }
      for (int k = 0; k < 3; ++k) t[k] = p[3 + k];
      break;
    case PP_MODEL_EULER: {  // ZXY order, ITK's default
      const double cx = std::cos(p[0]), sx = std::sin(p[0]), cy = std::cos(p[1]), sy = std::sin(p[1]), cz = std::cos(p[2]),
                   sz = std::sin(p[2]);
      const double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
      mat_mul(Rz, Rx, A);
      mat_mul(A, Ry, A);
      for (int k = 0; k < 3; ++k) t[k] = p[3 + k];
      break;
    }
  }
}

struct level_state {
  pp_ctx* ctx;
  const pp_linreg_level* L;
  const float* fixed;
  const float* moving;
  const int* fsize;
  const int* msize;
  const uint8_t* fmask;
  const uint8_t* mmask;
  int n;                  // parameters
  double Af[9], bf[3];    // virtual index -> fixed index
  double corners[8][3];   // physical corners of the virtual domain
  int evaluations;

  void setup() {
    mat_mul(L->f_p2i, L->v_i2p, Af);
    double d[3];
    for (int k = 0; k < 3; ++k) d[k] = L->v_origin[k] - L->f_origin[k];
    mat_vec(L->f_p2i, d, bf);
    int c = 0;
    for (int kz = 0; kz < 2; ++kz)
      for (int ky = 0; ky < 2; ++ky)
        for (int kx = 0; kx < 2; ++kx, ++c) {
          const double idx[3] = {kx ? L->vsize[0] - 1.0 : 0.0, ky ? L->vsize[1] - 1.0 : 0.0, kz ? L->vsize[2] - 1.0 : 0.0};
          mat_vec(L->v_i2p, idx, corners[c]);
          for (int k = 0; k < 3; ++k) corners[c][k] += L->v_origin[k];
        }
    evaluations = 0;
  }

  // (A, off) of initial o model(params): q = A p + off
  void total(const double* params, double* A, double* off) const {
    double Am[9], t[3], Ac[3], o[3];
    decode(L->model, params, Am, t);
    mat_vec(Am, L->center, Ac);
    for (int k = 0; k < 3; ++k) o[k] = t[k] + L->center[k] - Ac[k];
    mat_mul(L->init_matrix, Am, A);
    mat_vec(L->init_matrix, o, off);
    for (int k = 0; k < 3; ++k) off[k] += L->init_offset[k];
  }

  // virtual index -> moving index
  void index_map(const double* params, double* Am, double* bm) const {
    double A[9], off[3], q[3];
    total(params, A, off);
    mat_mul(L->m_p2i, A, Am);
    mat_mul(Am, L->v_i2p, Am);
    mat_vec(A, L->v_origin, q);
    for (int k = 0; k < 3; ++k) q[k] += off[k] - L->m_origin[k];
    mat_vec(L->m_p2i, q, bm);
  }

  // -> PP_OK, value and d value / d (Am row-major, bm); *overlap = 0 when no sample point is valid
  int raw(const double* params, double* value, double* g_idx, int* overlap) {
    double Am[9], bm[3];
    index_map(params, Am, bm);
    ++evaluations;
    *overlap = 1;
    if (L->metric == 0) {
      double r[14];
      const int rc = pp_meansq_affine_f32(ctx, fixed, fsize, moving, msize, Af, bf, Am, bm, L->vsize, L->stride, fmask, mmask, r);
      if (rc) return rc;
      if (r[1] <= 0) {
        *overlap = 0;
        return PP_OK;
      }
      *value = r[0] / r[1];
      for (int k = 0; k < 12; ++k) g_idx[k] = r[2 + k] / r[1];
      return PP_OK;
    }
    double r[42];
    const int rc = pp_corr_moments_affine_f32(ctx, fixed, fsize, moving, msize, Af, bf, Am, bm, L->vsize, L->stride, fmask, mmask, r);
    if (rc) return rc;
    const double cnt = r[0];
    if (cnt <= 0) {
      *overlap = 0;
      return PP_OK;
    }
    const double fbar = r[1] / cnt, mbar = r[2] / cnt;
    const double sff = r[3] - cnt * fbar * fbar, smm = r[4] - cnt * mbar * mbar, sfm = r[5] - cnt * fbar * mbar;
    if (sff <= 1e-300 || smm <= 1e-300) {
      *value = 0.0;
      for (int k = 0; k < 12; ++k) g_idx[k] = 0.0;
      return PP_OK;
    }
    *value = -(sfm * sfm) / (sff * smm);
    for (int k = 0; k < 12; ++k) {
      const double G = r[6 + k], FG = r[18 + k], MG = r[30 + k];
      const double dsfm = FG - fbar * G, dsmm = 2.0 * (MG - mbar * G);
      g_idx[k] = -(2.0 * sfm / (sff * smm) * dsfm - (sfm * sfm) / (sff * smm * smm) * dsmm);
    }
    return PP_OK;
  }

  // values only, for up to 16 parameter vectors (row-major [k][n]); +inf where nothing overlaps
  int values(int k, const double* plist, double* out) {
    double Am[16 * 9], bm[16 * 3], r[16 * 6];
    for (int c = 0; c < k; ++c) index_map(plist + (size_t)c * n, Am + c * 9, bm + c * 3);
    evaluations += k;
    const int rc = pp_metric_values_affine_f32(ctx, L->metric, fixed, fsize, moving, msize, Af, bf, k, Am, bm, L->vsize, L->stride, fmask,
                                               mmask, r);
    if (rc) return rc;
    for (int c = 0; c < k; ++c) {
      const double* row = r + c * 6;
      if (L->metric == 0) {
        out[c] = row[1] > 0 ? row[0] / row[1] : INF;
        continue;
      }
      const double cnt = row[0];
      if (cnt <= 0) {
        out[c] = INF;
        continue;
      }
      const double fbar = row[1] / cnt, mbar = row[2] / cnt;
      const double sff = row[3] - cnt * fbar * fbar, smm = row[4] - cnt * mbar * mbar, sfm = row[5] - cnt * fbar * mbar;
      out[c] = (sff <= 1e-300 || smm <= 1e-300) ? 0.0 : -(sfm * sfm) / (sff * smm);
    }
    return PP_OK;
  }

  // chain rule through params -> (Am, bm), by central differences of the (cheap, exact-to-1e-10) map
  int value_and_gradient(const double* params, double* value, double* grad, int* overlap) {
    double g_idx[12];
    const int rc = raw(params, value, g_idx, overlap);
    if (rc || !*overlap) return rc;
    std::vector<double> pp(params, params + n), pm(params, params + n);
    for (int i = 0; i < n; ++i) {
      const double h = 1e-6 * std::fmax(1.0, std::fabs(params[i]));
      pp[i] = params[i] + h;
      pm[i] = params[i] - h;
      double Ap[9], bp[3], An[9], bn[3];
      index_map(pp.data(), Ap, bp);
      index_map(pm.data(), An, bn);
      double acc = 0.0;
      for (int k = 0; k < 9; ++k) acc += g_idx[k] * ((Ap[k] - An[k]) / (2 * h));
      for (int k = 0; k < 3; ++k) acc += g_idx[9 + k] * ((bp[k] - bn[k]) / (2 * h));
      grad[i] = acc;
      pp[i] = pm[i] = params[i];
    }
    return PP_OK;
  }

  // itk::RegistrationParameterScalesFromPhysicalShift: largest corner displacement caused by a parameter change
  double max_shift(const double* params, const double* delta) const {
    std::vector<double> p1(n);
    update_parameters(L->model, n, params, delta, p1.data());   // ScalesFromShiftBase::ComputeSampleShifts -> UpdateTransformParameters
    double A0[9], o0[3], A1[9], o1[3];
    total(params, A0, o0);
    total(p1.data(), A1, o1);
    double dA[9], best = 0.0;
    for (int k = 0; k < 9; ++k) dA[k] = A1[k] - A0[k];
    for (int c = 0; c < 8; ++c) {
      double d[3];
      mat_vec(dA, corners[c], d);
      double s = 0.0;
      for (int k = 0; k < 3; ++k) {
        d[k] += o1[k] - o0[k];
        s += d[k] * d[k];
      }
      best = std::fmax(best, std::sqrt(s));
    }
    return best;
  }

  void scales(const double* params, double* s) const {
    const double variation = 0.01;
    std::vector<double> dlt(n, 0.0);
    double fill = INF;
    for (int i = 0; i < n; ++i) {
      dlt[i] = variation;
      s[i] = max_shift(params, dlt.data());
      dlt[i] = 0.0;
      if (s[i] > 1e-12) fill = std::fmin(fill, s[i]);
    }
    if (fill == INF) fill = 1.0;
    for (int i = 0; i < n; ++i) {
      if (s[i] <= 1e-12) s[i] = fill;
      s[i] = (s[i] * s[i]) / (variation * variation);
    }
  }

  double step_scale(const double* params, const double* step) const {
    const double variation = 0.01;
    double m = 0.0;
    for (int i = 0; i < n; ++i) m = std::fmax(m, std::fabs(step[i]));
    if (m <= 1e-300) return 0.0;
    const double factor = variation / m;
    std::vector<double> d(n);
    for (int i = 0; i < n; ++i) d[i] = step[i] * factor;
    return max_shift(params, d.data()) / factor;
  }
};

// itk::Function::WindowConvergenceMonitoringFunction::GetConvergenceValue: the last `window` energies, divided by the sum of
// their magnitudes, at t = i / (window - 1), approximated by BSplineScatteredDataPointSetToImageFilter with spline order 1, two
// control points and one level -- the Lee / Wolberg / Shin scattered-data update, NOT a least-squares line: with the hat weights
// w0 = 1 - t, w1 = t of a sample, control point k = sum(w_k^3 e / (w0^2 + w1^2)) / sum(w_k^2) -- and the value is minus the
// slope between the two control points.  The filter moves a sample on the domain's end (t = 1) inside by its B-spline epsilon
// (1e-3 of the 0.1 spacing of its parametric grid).  "Not yet" = +inf until the window is full.  ITK 5.3 from memory.
double window_convergence(const std::vector<double>& values, int window) {
  if ((int)values.size() < window) return INF;
  const double* e = values.data() + values.size() - window;
  double tot = 0.0;
  for (int i = 0; i < window; ++i) tot += std::fabs(e[i]);
  if (tot == 0.0) return 0.0;
  double delta[2] = {0.0, 0.0}, omega[2] = {0.0, 0.0};
  for (int i = 0; i < window; ++i) {
    double t = (double)i / (window - 1);
    if (std::fabs(t - 1.0) <= 1e-4) t = 1.0 - 1e-4;
    const double w[2] = {1.0 - t, t}, w2 = w[0] * w[0] + w[1] * w[1];
    for (int k = 0; k < 2; ++k) {
      delta[k] += (e[i] / tot) * (w[k] * w[k] * w[k] / w2);
      omega[k] += w[k] * w[k];
    }
  }
  return -(delta[1] / omega[1] - delta[0] / omega[0]);
}

// itk::GradientDescentLineSearchOptimizerv4::GoldenSectionSearch with speculative, batched probing.
struct golden_search {
  level_state* S;
  const double* base;  // parameters at learning rate 0
  const double* g;     // scaled gradient: params(e) = UpdateTransformParameters(base, -e g)
  double eps;
  int max_iter, depth;
  std::vector<double> kx, kv;  // probed learning rates and their values

  bool known(double x, double* v) const {
    for (size_t i = 0; i < kx.size(); ++i)
      if (kx[i] == x) {
        if (v) *v = kv[i];
        return true;
      }
    return false;
  }
  static double probe(double a, double b, double c) {
    const double resphi = 2.0 - (1.0 + std::sqrt(5.0)) / 2.0;
    return (c - b) > (b - a) ? b + resphi * (c - b) : b - resphi * (b - a);
  }
  static void children(double a, double b, double c, double x, double lo[3], double hi[3]) {
    if ((c - b) > (b - a)) {
      lo[0] = b, lo[1] = x, lo[2] = c;
      hi[0] = a, hi[1] = b, hi[2] = x;
    } else {
      lo[0] = a, lo[1] = x, lo[2] = b;
      hi[0] = x, hi[1] = b, hi[2] = c;
    }
  }
  void speculate(double a, double b, double c, int levels, int left, std::vector<double>& want) const {
    if (levels == 0 || left == 0) return;
    const double x = probe(a, b, c);
    if (std::fabs(c - a) < eps * (std::fabs(b) + std::fabs(x))) return;
    bool have = known(x, nullptr);
    for (size_t i = 0; i < want.size() && !have; ++i) have = want[i] == x;
    if (!have) want.push_back(x);
    double lo[3], hi[3];
    children(a, b, c, x, lo, hi);
    speculate(lo[0], lo[1], lo[2], levels - 1, left - 1, want);
    speculate(hi[0], hi[1], hi[2], levels - 1, left - 1, want);
  }
  int evaluate(const std::vector<double>& want) {
    const int n = S->n;
    std::vector<double> plist(want.size() * (size_t)n), vals(want.size());
    for (size_t off = 0; off < want.size(); off += 16) {
      const int k = (int)std::min<size_t>(16, want.size() - off);
      std::vector<double> step(n);
      for (int c = 0; c < k; ++c) {
        for (int i = 0; i < n; ++i) step[i] = -want[off + c] * g[i];
        update_parameters(S->L->model, n, base, step.data(), plist.data() + (size_t)c * n);
      }
      const int rc = S->values(k, plist.data(), vals.data() + off);
      if (rc) return rc;
    }
    for (size_t i = 0; i < want.size(); ++i) {
      kx.push_back(want[i]);
      kv.push_back(vals[i]);
    }
    return PP_OK;
  }
  int run(double a, double b, double c, double* result) {
    bool have_fb = false;
    double fb = 0.0;
    for (int it = 0; it < max_iter; ++it) {
      const double x = probe(a, b, c);
      if (std::fabs(c - a) < eps * (std::fabs(b) + std::fabs(x))) {
        *result = (c + a) / 2.0;
        return PP_OK;
      }
      if (!known(x, nullptr) || (!have_fb && !known(b, nullptr))) {
        std::vector<double> want;
        speculate(a, b, c, depth, max_iter - it, want);
        if (!have_fb && !known(b, nullptr)) {
          bool in = false;
          for (double w : want) in = in || w == b;
          if (!in) want.push_back(b);
        }
        const int rc = evaluate(want);
        if (rc) return rc;
      }
      double fx = 0.0;
      known(x, &fx);
      if (!have_fb) {
        known(b, &fb);
        have_fb = true;
      }
      double lo[3], hi[3];
      children(a, b, c, x, lo, hi);
      if (fx < fb) {
        a = lo[0], b = lo[1], c = lo[2];
        fb = fx;
      } else {
        a = hi[0], b = hi[1], c = hi[2];
      }
    }
    *result = (c + a) / 2.0;
    return PP_OK;
  }
};

}  // namespace

extern "C" int pp_linear_num_parameters(int model) { return model_params(model); }

extern "C" int pp_linear_optimize_f32(pp_ctx* ctx, const float* fixed, const int fsize[3], const float* moving, const int msize[3],
                                      const uint8_t* fixed_mask, const uint8_t* moving_mask, const pp_linreg_level* level, double* params,
                                      pp_linreg_stats* stats, double* history, int history_capacity) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_REQUIRE(ctx, fixed && fsize && moving && msize && level && params, "pp_linear_optimize_f32: NULL argument");
  const int n = model_params(level->model);
  PP_REQUIRE(ctx, n > 0, "pp_linear_optimize_f32: unknown transform model");
  PP_REQUIRE(ctx, level->metric == 0 || level->metric == 1, "pp_linear_optimize_f32: metric must be 0 (mean squares) or 1 (correlation)");
  PP_REQUIRE(ctx, level->optimizer == PP_OPT_GD || level->optimizer == PP_OPT_GD_LINE_SEARCH, "pp_linear_optimize_f32: unknown optimiser");
  PP_REQUIRE(ctx, level->iterations >= 0 && level->stride >= 1, "pp_linear_optimize_f32: bad iteration count or stride");
  PP_REQUIRE(ctx, level->vsize[0] >= 1 && level->vsize[1] >= 1 && level->vsize[2] >= 1, "pp_linear_optimize_f32: empty virtual domain");
  level_state S{ctx, level, fixed, moving, fsize, msize, fixed_mask, moving_mask, n, {}, {}, {}, 0};
  S.setup();
  const int depth = level->speculation < 1 ? 1 : (level->speculation > 4 ? 4 : level->speculation);
  pp_fsamp_scope fixed_samples_scope(ctx);   // the fixed image is constant for this call: its lattice samples are evaluated once

  const bool return_best = (level->flags & PP_LINREG_RETURN_BEST) != 0;
  std::vector<double> p(params, params + n), best(p), prev(p), scales(n), grad(n), g(n), step(n), hist;
  S.scales(p.data(), scales.data());
  double learning_rate = 1.0, best_value = INF, last_value = INF;
  int stop = PP_LINREG_STOP_ITERATIONS, done = 0;
  for (int it = 0; it < level->iterations; ++it) {
    double value = 0.0;
    int overlap = 1;
    int rc = S.value_and_gradient(p.data(), &value, grad.data(), &overlap);
    if (rc) return rc;
    if (!overlap) {
      if (it == 0) return pp_fail(ctx, PP_ERR_NO_OVERLAP, "linear registration: no valid sample points (images do not overlap)");
      // stepped off the overlap (ITK's metric would warn, return its maximum and a zero derivative, and the optimiser would sit
      // there): the level ends at the last point that had samples
      stop = PP_LINREG_STOP_NO_OVERLAP;
      p = prev;
      break;
    }
    last_value = value;
    if (value < best_value) {
      best_value = value;
      best = p;
    }
    hist.push_back(value);
    if (history && it < history_capacity) history[it] = value;
    done = it + 1;
    if (window_convergence(hist, 10) <= 1e-6) {
      stop = PP_LINREG_STOP_CONVERGED;
      break;
    }
    for (int i = 0; i < n; ++i) g[i] = grad[i] / scales[i];  // ModifyGradientByScales
    if (it == 0) {                                         // estimateLearningRate = Once (per StartOptimization, i.e. per level)
      std::vector<double> neg(n);
      for (int i = 0; i < n; ++i) neg[i] = -g[i];
      const double ss = S.step_scale(p.data(), neg.data());
      learning_rate = ss > std::numeric_limits<double>::epsilon() ? level->v_min_spacing / ss : 1.0;
    }
    if (level->optimizer == PP_OPT_GD_LINE_SEARCH) {
      golden_search gs{&S, p.data(), g.data(), 0.01, 20, depth, {}, {}};
      double lr = 0.0;
      rc = gs.run(0.0, learning_rate, 5.0 * learning_rate, &lr);
      if (rc) return rc;
      if (lr > 0) learning_rate = lr;
    }
    prev = p;
    for (int i = 0; i < n; ++i) step[i] = -learning_rate * g[i];
    update_parameters(level->model, n, prev.data(), step.data(), p.data());   // m_Metric->UpdateTransformParameters(m_Gradient)
  }
  double last = last_value;       // GetMetricValue(): the value of the last evaluation, one step behind the returned point
  if (return_best) {              // SetOptimizerAsGradientDescent(..., returnBestParametersAndValue=True): best point evaluated
    const int rc = S.values(1, p.data(), &last);
    if (rc) return rc;
    if (last > best_value) {
      p = best;
      last = best_value;
    }
  }
  for (int i = 0; i < n; ++i) params[i] = p[i];
  if (stats) {
    stats->iterations = done;
    stats->evaluations = S.evaluations;
    stats->stop = stop;
    stats->value = last;
    stats->learning_rate = learning_rate;
  }
  return PP_OK;
}
