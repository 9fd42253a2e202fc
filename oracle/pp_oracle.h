/*
 * oracle/pp_oracle.h -- CPU restatement of the reference's deformable-registration /
 * label-fusion hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under platipy_amd/ may include, link, load or call
 * this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * there only as the checker / reported CPU baseline, never as the thing measured or shipped.
 *
 * PARITY UNPINNED.  The reference (pyplati/platipy) performs every voxel-level operation on
 * this path by calling SimpleITK 2.3.1 (ITK 5.3), a third-party wheel that is not vendored
 * under /root/reference and is not installed in the build container (no network).  The
 * reference's own tests hold no golden vectors for this path (only Dice > 0.99 on synthetic
 * spheres, platipy/imaging/tests/test_cardiac.py:142).  The functions below restate the
 * published ITK 5.3 algorithms that the reference's call sites invoke; each cites the
 * reference call site (file:line under /root/reference) and the ITK class it restates.
 * tools/compare_with_sitk.py re-checks every stage against SimpleITK wherever it exists.
 *
 * Conventions: volumes are [Z][Y][X] (x fastest), size = {nx, ny, nz}; vector fields are
 * planar, [3][Z][Y][X] with component 0 = x displacement in mm (physical units).
 * Displacement fields are fp64 as in the reference (sitkVectorFloat64, deformable.py:97).
 */
#ifndef PP_ORACLE_H
#define PP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int size[3];         /* nx, ny, nz */
  double spacing[3];   /* mm */
  double origin[3];    /* mm */
  double direction[9]; /* row-major 3x3 */
} orc_geom;

typedef struct {
  double metric;      /* mean squared intensity difference, ESM m_Metric     */
  double rms_change;  /* sqrt(mean |U|^2) of the raw update, ESM m_RMSChange */
  double sum_sq_diff;
  double sum_sq_change;
  int64_t n_pixels;
  int elapsed_iterations;
} orc_demons_stats;

enum { ORC_INTERP_NEAREST = 1, ORC_INTERP_LINEAR = 2 };

/* itkGaussianOperator::GenerateCoefficients (used by every Gaussian FIR on the path).
 * coeffs receives 2*radius+1 taps (left edge .. centre .. right edge); returns radius, or
 * -1 if cap is too small. */
int orc_gaussian_operator(double variance, double max_error, int max_kernel_width,
                          double* coeffs, int cap);

/* itkDiscreteGaussianImageFilter on a float image (fusion.py:168,279; registration/utils.py:226).
 * variance in physical units when use_spacing != 0.  Pass order z,y,x; fp64 accumulate,
 * fp32 intermediate images; ZeroFluxNeumann edges. */
int orc_discrete_gaussian_f32(const float* in, float* out, const int size[3],
                              const double spacing[3], const double variance[3],
                              double max_error, int max_kernel_width, int use_spacing);

/* itkPDEDeformableRegistrationFilter::SmoothDisplacementField / SmoothUpdateField
 * (deformable.py:248-257 configure it).  sigma in voxels; in-place on fp64 planar field. */
int orc_smooth_field_f64(double* field, const int size[3], const double sigma_vox[3],
                         double max_error, int max_kernel_width);

/* itkWarpImageFilter as used inside itkESMDemonsRegistrationFunction::InitializeIteration:
 * out(x) = M(x + D(x)) trilinear, edge_value where x + D(x) leaves M's buffer. */
int orc_warp_image_f32(const float* moving, const orc_geom* gm, const double* field,
                       const orc_geom* gout, float edge_value, float* out);

/* itkESMDemonsRegistrationFunction::ComputeUpdate over the whole image (symmetric gradient).
 * update is fp64 planar [3][Z][Y][X]. */
int orc_esm_update(const float* fixed, const float* warped, const orc_geom* g,
                   double max_step_length, double intensity_threshold,
                   double denominator_threshold, double* update, orc_demons_stats* stats);

/* sitk.FastSymmetricForcesDemonsRegistrationFilter.Execute(fixed, moving)
 * (deformable.py:149, configured at :244-257).  field (out) starts from zero. */
int orc_demons_execute(const float* fixed, const float* moving, const orc_geom* g,
                       int n_iterations, const double sigma_d_vox[3], const double sigma_u_vox[3],
                       int smooth_displacement, int smooth_update, double max_rms_error,
                       double max_step_length, double intensity_threshold, double max_error,
                       int max_kernel_width, double* field, orc_demons_stats* stats);

/* itkResampleImageFilter, scalar float image.  Transform: q = p (identity), or
 * q = A p + t (affine, A row-major 3x3; pass NULL for identity), then optionally
 * q += D(q_in) with D a displacement field on geometry gd (itkDisplacementFieldTransform;
 * pass field = NULL for none).  Only one of affine / field is normally given. */
int orc_resample_f32(const float* in, const orc_geom* gin, const orc_geom* gout,
                     const double* affine_A, const double* affine_t, const double* field,
                     const orc_geom* gd, int interp, double default_value, float* out);
int orc_resample_u8(const uint8_t* in, const orc_geom* gin, const orc_geom* gout,
                    const double* affine_A, const double* affine_t, const double* field,
                    const orc_geom* gd, int interp, double default_value, uint8_t* out);
/* sitk.Resample on a vector fp64 image (deformable.py:130,137,154,185): identity transform or
 * displacement-field transform, linear interpolation, default 0. */
int orc_resample_vec_f64(const double* in, const orc_geom* gin, const orc_geom* gout,
                         const double* field, const orc_geom* gd, double* out);

/* itkSmoothingRecursiveGaussianImageFilter on a vector fp64 image (deformable.py:157-158).
 * sigma in physical units (the reference passes voxel-unit numbers there -- quirk N2). */
int orc_recursive_gaussian_vec_f64(double* field, const orc_geom* g, const double sigma[3]);
int orc_recursive_gaussian_f32(const float* in, float* out, const orc_geom* g,
                               const double sigma[3]);
/* one directional pass (order 0 / 1) of itk::RecursiveGaussianImageFilter: the parts of GradientRecursiveGaussianImageFilter */
int orc_recursive_gaussian_pass_f32(const float* in, float* out, const orc_geom* g, int axis, double sigma,
                                    int order, int normalize_across_scale);

/* fusion.py:148-169 (vote_type "local"):  w = 1 / (DiscreteGaussian((T-M)^2, sigma^2) + eps). */
int orc_weight_map_local(const float* target, const float* moving, const int size[3],
                         const double spacing[3], double sigma, double epsilon, float* weight);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
