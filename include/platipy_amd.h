/*
 * include/platipy_amd.h -- C ABI of libplatipy_hip.so (gfx950 / MI355X).
 *
 * The reference (pyplati/platipy) has no C/FFI boundary on this path: its L2 Python functions
 * call SimpleITK (SWIG -> ITK C++) directly.  This header is therefore the boundary a
 * maintainer would bind *instead of* those SimpleITK calls; every entry point names the
 * reference call site (file:line under the reference tree) whose SimpleITK call it replaces.
 * platipy_amd/_lib.py is the ctypes binding; INTEGRATION.md shows the reference-side stub.
 *
 * Rules of the ABI
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - All volume pointers are caller-owned DEVICE pointers (e.g. torch tensor data_ptr());
 *    the library never frees or retains them past the call.  Scratch memory belongs to the
 *    ctx and grows on demand (hipMalloc only when a call needs more than any earlier one).
 *  - Every call enqueues on the ctx's stream and returns; pp_sync() waits.  The one
 *    exception is a call given a non-NULL host `stats` pointer, which synchronises the
 *    stream before returning so the statistics are valid.
 *  - Return value: PP_OK (0) or a negative pp_status; pp_last_error(ctx) describes the last
 *    failure.  Nothing throws across the ABI and nothing calls exit().
 *  - One ctx per (device, stream); no global state, so N ctxs can drive N GPUs/streams.
 *
 * Layout: scalar volumes are [Z][Y][X] (x fastest), size = {nx, ny, nz}.  Displacement
 * fields are planar fp32, [3][Z][Y][X]; plane c holds the c-th physical (mm) component.
 */
#ifndef PLATIPY_AMD_H
#define PLATIPY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_ABI_VERSION 2

typedef enum {
  PP_OK = 0,
  PP_ERR_ARG = -1,         /* NULL / inconsistent argument                       */
  PP_ERR_HIP = -2,         /* a HIP runtime call or kernel launch failed          */
  PP_ERR_ALLOC = -3,       /* workspace allocation failed                         */
  PP_ERR_UNSUPPORTED = -4, /* valid request this build does not implement         */
  PP_ERR_SIZE = -5,        /* volume too small / too large for the operation      */
  PP_ERR_NO_OVERLAP = -6   /* linear registration: no valid sample point at start */
} pp_status;

enum { PP_INTERP_NEAREST = 1, PP_INTERP_LINEAR = 2, PP_INTERP_BSPLINE = 3 }; /* = sitk.sitkNearestNeighbor / sitkLinear / sitkBSpline */
enum { PP_MORPH_DILATE = 0, PP_MORPH_ERODE = 1, PP_MORPH_CLOSE = 2 };
enum { PP_DTYPE_U8 = 0, PP_DTYPE_F32 = 1 };

enum { PP_DEMONS_AUTO = 0, PP_DEMONS_STAGED = 1, PP_DEMONS_FUSED = 2 };

typedef struct pp_ctx pp_ctx;

typedef struct {
  int size[3];         /* nx, ny, nz */
  double spacing[3];   /* mm         */
  double origin[3];    /* mm         */
  double direction[9]; /* row-major  */
} pp_geom;

/* Parameters of sitk.FastSymmetricForcesDemonsRegistrationFilter as the reference
 * configures it (registration/deformable.py:244-257); pp_demons_default_params() fills
 * SimpleITK 2.3.1's defaults. */
typedef struct {
  int iterations;               /* SetNumberOfIterations (deformable.py:144)              */
  double sigma_d_vox[3];        /* SetStandardDeviations, voxels (deformable.py:253-257)  */
  double sigma_u_vox[3];        /* UpdateFieldStandardDeviations, voxels (default 1.0)    */
  int smooth_displacement;      /* SetSmoothDisplacementField (deformable.py:249)         */
  int smooth_update;            /* SetSmoothUpdateField (deformable.py:248)               */
  double max_rms_error;         /* MaximumRMSError, 0.02; <= 0 disables the early halt    */
  double max_step_length;       /* MaximumUpdateStepLength, 0.5                           */
  double intensity_threshold;   /* IntensityDifferenceThreshold, 0.001                    */
  double denominator_threshold; /* ESM m_DenominatorThreshold, 1e-9                       */
  double max_error;             /* GaussianOperator MaximumError, 0.1                     */
  int max_kernel_width;         /* GaussianOperator MaximumKernelWidth, 30                */
  int variant;                  /* PP_DEMONS_AUTO | _STAGED | _FUSED                      */
} pp_demons_params;

typedef struct {
  double metric;         /* GetMetric(): mean squared intensity difference            */
  double rms_change;     /* GetRMSChange(): sqrt(mean |update|^2), raw update         */
  double sum_sq_diff;
  double sum_sq_change;
  int64_t n_pixels;
  int elapsed_iterations; /* GetElapsedIterations()                                   */
  int halted;             /* 1 if the RMS rule stopped the loop early                 */
} pp_demons_stats;

/* ---- context --------------------------------------------------------------------- */
int pp_abi_version(void);
/* The PP_* measurement / debugging switches are read from the process environment ONCE, when the library
 * first needs one; launch paths never call getenv.  pp_reload_switches() takes the snapshot again -- for tests and A/B tools
 * that flip a switch inside one process; not while another thread is inside the library.  (No reference counterpart.) */
void pp_reload_switches(void);
int pp_create(int device, void* hip_stream, pp_ctx** out);
void pp_destroy(pp_ctx* ctx);
const char* pp_last_error(const pp_ctx* ctx);
int pp_set_stream(pp_ctx* ctx, void* hip_stream);
int pp_sync(pp_ctx* ctx);
size_t pp_workspace_bytes(const pp_ctx* ctx);

/* Optional per-kernel timing with HIP events recorded on the ctx stream around every kernel
 * launch of the demons loop (bench.py's roofline figures).  Off by default: when off no event
 * is created or recorded.  `on` = 1 brackets every launch, `on` = k > 1 every k-th launch of each
 * kernel (a pair of events costs the stream ~7 us); `launches` counts the bracketed ones.
 * pp_profile_read synchronises, returns the number of distinct kernels (<= cap entries written)
 * and resets the accumulators. */
typedef struct {
  char name[48];
  int launches;
  double total_ms;
} pp_profile_entry;
int pp_profile_enable(pp_ctx* ctx, int on);
int pp_profile_read(pp_ctx* ctx, pp_profile_entry* out, int cap);

/* ---- host helpers ---------------------------------------------------------------- */
/* itk::GaussianOperator coefficients (every FIR below uses them).  taps gets 2r+1 values,
 * returns r or a negative pp_status. */
int pp_gauss_taps(double variance, double max_error, int max_kernel_width, double* taps, int cap);
void pp_demons_default_params(pp_demons_params* p);

/* ---- Gaussian FIR ------------------------------------------------------------------ */
/* sitk.DiscreteGaussian(image, variance, maximumKernelWidth, maximumError=0.01,
 * useImageSpacing=True): registration/utils.py:226, label/fusion.py:168,279. */
int pp_discrete_gaussian_f32(pp_ctx* ctx, const float* in, float* out, const int size[3],
                             const double spacing[3], const double variance[3], double max_error,
                             int max_kernel_width, int use_image_spacing);
/* The same filter when only rows (y, z) with need_y[y] && need_z[z] (device uint8 masks of ny / nz entries) of the
 * result will be read -- the pyramid's blur is followed by a resample onto a coarser grid (registration/utils.py:226,
 * :257-267).  Values that are produced equal the dense filter's bit for bit; other entries of out are unspecified. */
int pp_discrete_gaussian_rows_f32(pp_ctx* ctx, const float* in, float* out, const int size[3],
                                  const double spacing[3], const double variance[3], double max_error,
                                  int max_kernel_width, int use_image_spacing, const uint8_t* need_y,
                                  const uint8_t* need_z);
/* PDEDeformableRegistrationFilter::SmoothDisplacementField / SmoothUpdateField, in place on
 * a planar 3-vector field; sigma in voxels (deformable.py:248-257). */
int pp_smooth_field_f32(pp_ctx* ctx, float* field, const int size[3], const double sigma_vox[3],
                        double max_error, int max_kernel_width);

/* ---- recursive (IIR) Gaussian ------------------------------------------------------ */
/* sitk.SmoothingRecursiveGaussian(dvf_total, sigma) (deformable.py:157-158), in place on a
 * planar field; sigma in the units ITK reads them in (mm). */
int pp_recursive_gaussian_field_f32(pp_ctx* ctx, float* field, const pp_geom* g,
                                    const double sigma[3]);
int pp_recursive_gaussian_f32(pp_ctx* ctx, const float* in, float* out, const pp_geom* g,
                              const double sigma[3]);

/* ONE directional pass of itk::RecursiveGaussianImageFilter over a scalar volume (in != out): order 0 the Gaussian, order 1 its
 * first derivative along `axis` per VOXEL (a unit ramp answers 1; times sigma when normalize_across_scale, as ITK's
 * NormalizeAcrossScale).  The building block of itk::GradientRecursiveGaussianImageFilter -- derivative along one axis, then
 * Gaussians along the others -- which itk::ImageToImageMetricv4 runs over the moving image (sigma = its largest spacing) inside
 * registration.Execute (registration/linear.py:238) as the metric's default gradient source; the host chains the passes
 * (platipy_amd/registration/linear.py, itk_sampling=True). */
int pp_recursive_gaussian_pass_f32(pp_ctx* ctx, const float* in, float* out, const pp_geom* g, int axis, double sigma, int order,
                                   int normalize_across_scale);

/* ---- warp / resample --------------------------------------------------------------- */
/* out(x) = moving(x + D(x)), moving/field/out on one grid: itk::WarpImageFilter inside the
 * demons loop (edge = FLT_MAX sentinel) and sitk.Resample(m_image, tfm_total, interp)
 * (deformable.py:140, edge 0) / the final warp (deformable.py:281-301). */
int pp_warp_f32(pp_ctx* ctx, const float* moving, const float* field, const pp_geom* g,
                float edge_value, float* out);
/* sitk.ResampleImageFilter (registration/utils.py:176-190, :257-267): out grid gout, input
 * grid gin, transform q = A p + t (NULL = identity) followed by q += D(p) for a field D
 * sampled on gout (NULL = none).  Coordinates are computed in fp64. */
int pp_resample_f32(pp_ctx* ctx, const float* in, const pp_geom* gin, const pp_geom* gout,
                    const double* affine_A, const double* affine_t, const float* field,
                    int interp, double default_value, float* out);
int pp_resample_u8(pp_ctx* ctx, const uint8_t* in, const pp_geom* gin, const pp_geom* gout,
                   const double* affine_A, const double* affine_t, const float* field,
                   int interp, double default_value, uint8_t* out);
/* itk::BSplineDecompositionImageFilter (spline order 3): samples -> B-spline coefficients, mirror boundaries; `out` may be
 * `in`.  pp_resample_f32 with interp = PP_INTERP_BSPLINE expects this coefficient volume as its input
 * (itk::BSplineInterpolateImageFunction: any sitk interpolator may reach registration/utils.py:176-190). */
int pp_bspline_prefilter_f32(pp_ctx* ctx, const float* in, const int size[3], float* out);
/* sitk.Resample on the vector field itself (deformable.py:130,137,185): linear, default 0. */
int pp_resample_field_f32(pp_ctx* ctx, const float* in, const pp_geom* gin, const pp_geom* gout,
                          float* out);
/* sitk.TransformToDisplacementField(initial_transform, sitkVectorFloat64, fixed grid) (deformable.py:101-108) for a
 * linear transform q = A p + t (fp64 coordinates): out(idx) = (A - I) p(idx) + t, plus `add_field` (planar, on the
 * same grid, may be NULL) -- the displacement part of a composite whose last member is a displacement field. */
int pp_transform_to_field_f32(pp_ctx* ctx, const pp_geom* g, const double* affine_A, const double* affine_t,
                              const float* add_field, float* out);
/* dvf_total + sitk.Resample(dvf_iter, DisplacementFieldTransform(dvf_total))
 * (deformable.py:154): total(x) += iter(x + total(x)), 0 outside; both on grid g. */
int pp_compose_field_f32(pp_ctx* ctx, float* total, const float* iter, const pp_geom* g);

/* ---- demons ------------------------------------------------------------------------ */
/* One itk::ESMDemonsRegistrationFunction::ComputeUpdate sweep (symmetric gradient) over
 * the grid: update = planar field.  stats may be NULL. */
int pp_demons_force_f32(pp_ctx* ctx, const float* fixed, const float* warped, const pp_geom* g,
                        const pp_demons_params* p, float* update, pp_demons_stats* stats);
/* registration_algorithm.Execute(f_image, m_image) (deformable.py:149): the whole inner
 * loop on device, field starting from zero.  stats may be NULL (then fully asynchronous). */
int pp_demons_execute_f32(pp_ctx* ctx, const float* fixed, const float* moving, const pp_geom* g,
                          const pp_demons_params* p, float* field, pp_demons_stats* stats);
/* What an sitkIterationEvent observer of the filter reads after every iteration -- registration_method.AddCommand(
 * sitk.sitkIterationEvent, ...) printing GetElapsedIterations() / GetMetric() (deformable.py:260-264,
 * registration/utils.py:36-41).  The loop runs on the device without host round trips, so the per-iteration values are kept
 * in a device ring by the kernel that closes each iteration and read back here: entry k = GetMetric() / GetRMSChange() after
 * iteration k + 1 of the LAST pp_demons_execute_f32 on this ctx.  Returns the number of iterations that ran, or a negative
 * pp_status; entries written: min(that, cap, PP_DEMONS_HISTORY_CAPACITY = 4096 -- the ring keeps the first 4096 iterations of an
 * Execute, a return value above it says the later ones were not recorded).  Synchronises the stream. */
#define PP_DEMONS_HISTORY_CAPACITY 4096
int pp_demons_history(pp_ctx* ctx, double* metric, double* rms_change, int cap);

/* ---- label fusion ------------------------------------------------------------------ */
/* compute_weight_map(vote_type="local") (label/fusion.py:148-169):
 * w = 1 / (DiscreteGaussian((T - M)^2, sigma^2) + epsilon). */
int pp_weight_map_local_f32(pp_ctx* ctx, const float* target, const float* moving,
                            const int size[3], const double spacing[3], double sigma,
                            double epsilon, float* weight);
/* compute_weight_map(vote_type="block") (label/fusion.py:179-190):
 * w = factor * BoxMean((T - M)^2, radius)^(-|gain / 2|), box mean with ZeroFluxNeumann edges; radius in voxels (x, y, z). */
int pp_weight_map_block_f32(pp_ctx* ctx, const float* target, const float* moving, const int size[3],
                            const int radius[3], double factor, double gain, float* weight);
/* sum of squared differences (vote_type="global", label/fusion.py:154-161), fp64 on host. */
int pp_sum_sq_diff_f32(pp_ctx* ctx, const float* a, const float* b, size_t n, double* result);
/* combine_labels accumulation (label/fusion.py:263,269-276), one atlas at a time:
 * wsum += w (if wsum != NULL);  wlsum += w * label. */
int pp_fuse_accumulate_u8(pp_ctx* ctx, const float* weight, const uint8_t* label, float* wsum,
                          float* wlsum, size_t n);
/* the same with a float label (probabilistic atlas labels: the reference casts every label to sitkFloat32 before
 * weighting, label/fusion.py:269-272, so values other than 0/1 are weighted as they are) */
int pp_fuse_accumulate_f32(pp_ctx* ctx, const float* weight, const float* label, float* wsum,
                           float* wlsum, size_t n);
/* P = wlsum / (wsum == 0 ? 1 : wsum)  (label/fusion.py:264-276) */
int pp_fuse_divide_f32(pp_ctx* ctx, const float* wlsum, const float* wsum, float* out, size_t n);
/* global min / max (RescaleIntensity, label/fusion.py:282; process_probability_image :305) */
int pp_minmax_f32(pp_ctx* ctx, const float* in, size_t n, float* min_out, float* max_out);
/* RescaleIntensity(0,1) given (min,max) then Threshold(lower, upper=1, outside=0)
 * (label/fusion.py:282-288), in place. */
int pp_rescale_threshold_f32(pp_ctx* ctx, float* data, size_t n, float in_min, float in_max,
                             float lower);
/* BinaryThreshold(prob / max_value >= threshold) -> uint8 (label/fusion.py:305-308) */
int pp_binary_threshold_f32(pp_ctx* ctx, const float* prob, size_t n, double max_value,
                            double threshold, uint8_t* out);

/* BinaryFillhole -> ConnectedComponent -> keep the largest component (label/fusion.py:310-328), face
 * connectivity.  `in`/`out` are uint8 masks (non-zero = foreground); fill_holes = 0 skips the hole filling.
 * If there is no foreground the (filled) input comes back, as the reference returns its binary image.
 * component_voxels (host, may be NULL) receives the size of the kept component; non-NULL synchronises. */
int pp_fillhole_largest_component_u8(pp_ctx* ctx, const uint8_t* in, const int size[3], int fill_holes,
                                     uint8_t* out, int64_t* component_voxels);

/* sitk.BinaryDilate / BinaryErode / BinaryMorphologicalClosing(mask, radius) with SimpleITK's defaults
 * (registration/utils.py:328-329; projects/multiatlas/run.py:421-423, projects/cardiac/run.py:1127-1129):
 * kernel = ITK ball, offset d in the element when sum_i (d_i / (radius_i + 0.5))^2 <= 1; dilation sees background
 * and erosion foreground outside the buffer; PP_MORPH_CLOSE = dilate then erode with a safe border (as if padded
 * by the radius).  Masks are 0 / non-zero in, 0 / 1 out; radius in voxels per axis (x, y, z), each <= 15. */
int pp_binary_morph_ball_u8(pp_ctx* ctx, const uint8_t* in, const int size[3], const int radius[3], int op,
                            uint8_t* out);

/* Bounding box of the voxels > 0 (label_to_roi, utils/crop.py:24-60): box (host) = {xmin, xmax, ymin, ymax, zmin,
 * zmax}; an empty volume gives xmin > xmax.  dtype = PP_DTYPE_U8 or PP_DTYPE_F32.  Synchronises. */
int pp_bounding_box(pp_ctx* ctx, const void* data, int dtype, const int size[3], int box[6]);

/* ---- iterative atlas removal ------------------------------------------------------- */
/* sitk.LabelContour(mask) with face connectivity (label/projection.py:85): object voxels that have a face
 * neighbour of a different value. */
int pp_label_contour_u8(pp_ctx* ctx, const uint8_t* mask, const int size[3], uint8_t* out);
/* sitk.SignedMaurerDistanceMap(mask, squaredDistance=False, useImageSpacing=True) (label/projection.py:80-82,
 * registration/utils.py:288-293): exact Euclidean distance (mm) to the nearest border voxel of the object
 * (object voxel with background in its 26-neighbourhood), 0 on the border; want_signed = 0 gives the absolute
 * map, otherwise inside is negative unless inside_positive. */
int pp_distance_map_f32(pp_ctx* ctx, const uint8_t* mask, const pp_geom* g, int want_signed,
                        int inside_positive, float* out);

/* ---- linear registration ----------------------------------------------------------- */
/* One evaluation of the mean-squares metric (itk::MeanSquaresImageToImageMetricv4, selected at
 * registration/linear.py:141-148, evaluated inside registration.Execute at :238) and its gradient
 * with respect to an affine map in INDEX space.  Sample points: every `stride`-th voxel, raster
 * order, of a virtual grid vsize (REGULAR sampling, linear.py:152-153).  For virtual index v:
 * f = trilinear(fixed, Af v + bf), m = trilinear(moving, Am v + bm); samples leaving either buffer or
 * rejected by a mask (nearest voxel == 0) are skipped.  result (host, 14 doubles):
 * [0] sum (f-m)^2, [1] count, [2..10] d/dAm (row-major), [11..13] d/dbm.  Synchronises. */
int pp_meansq_affine_f32(pp_ctx* ctx, const float* fixed, const int fsize[3], const float* moving,
                         const int msize[3], const double Af[9], const double bf[3],
                         const double Am[9], const double bm[3], const int vsize[3], int stride,
                         const uint8_t* fixed_mask, const uint8_t* moving_mask, double* result);

/* The same sampling for the correlation metric (itk::CorrelationImageToImageMetricv4, linear.py:142-143): raw
 * moments from which value and gradient of -corr^2 follow on the host.  result (host, 42 doubles):
 * [0] count [1] sum f [2] sum m [3] sum f^2 [4] sum m^2 [5] sum f m, then three blocks of 12 (d/dAm row-major 9,
 * d/dbm 3): [6..17] sum g, [18..29] sum f g, [30..41] sum m g, with g the interpolant-gradient terms g_r v_q / g_r. */
int pp_corr_moments_affine_f32(pp_ctx* ctx, const float* fixed, const int fsize[3], const float* moving,
                               const int msize[3], const double Af[9], const double bf[3],
                               const double Am[9], const double bm[3], const int vsize[3], int stride,
                               const uint8_t* fixed_mask, const uint8_t* moving_mask, double* result);

/* Line-search evaluations (ITK GradientDescentLineSearchOptimizerv4::GoldenSectionSearch inside
 * registration.Execute, registration/linear.py:238): metric VALUES only, for `ncand` (1..16) candidate moving
 * maps Am[c][9], bm[c][3] in one launch -- the host speculates the next levels of the search tree.  Sampling,
 * validity and interpolation exactly as pp_meansq_affine_f32 / pp_corr_moments_affine_f32.  result (host,
 * ncand x 6 doubles): metric 0 -> [sum (f-m)^2, count, -, -, -, -]; metric 1 -> [count, sum f, sum m, sum f^2,
 * sum m^2, sum f m].  A candidate's numbers do not depend on the others in the call.  Synchronises. */
int pp_metric_values_affine_f32(pp_ctx* ctx, int metric, const float* fixed, const int fsize[3],
                                const float* moving, const int msize[3], const double Af[9],
                                const double bf[3], int ncand, const double* Am, const double* bm,
                                const int vsize[3], int stride, const uint8_t* fixed_mask,
                                const uint8_t* moving_mask, double* result);

/* ITK's sample-point jitter for every metric entry point of this section (registration.SetMetricSamplingPercentage(rate,
 * seed=42) + SetMetricSamplingStrategy(REGULAR), registration/linear.py:151-152): itk::ImageRegistrationMethodv4 perturbs each
 * REGULAR sample point by a seeded normal variate times a third of the virtual spacing per axis.  `jitter` (device, caller-owned,
 * must stay valid until replaced): 3 floats per sample of the raster walk, in VIRTUAL-INDEX units, added to the sample's lattice
 * index before the fixed and moving maps are applied; `nsamples` its length in samples (>= the lattice's sample count of every
 * later call, else that call fails).  NULL / 0 restores the plain lattice (the default).  The host draws the variates
 * (platipy_amd/registration/linear.py, itk_sampling=True). */
int pp_linear_set_sample_jitter(pp_ctx* ctx, const float* jitter, size_t nsamples);

/* ITK's moving-image gradient source for the gradient-bearing metric entry points of this section (pp_meansq_affine_f32,
 * pp_corr_moments_affine_f32, pp_mi_gradient_f32, pp_linear_optimize_f32): itk::ImageToImageMetricv4 (inside
 * registration.Execute, registration/linear.py:238) by default does not differentiate the intensity interpolant but LINEARLY
 * interpolates a gradient image it computes once per level with itk::GradientRecursiveGaussianImageFilter (sigma = the moving
 * image's largest spacing).  `gradient` (device, caller-owned, valid until replaced): that image as three volumes
 * [3][Z][Y][X] of the moving image's size `msize`, converted to moving-INDEX units (d intensity / d index); NULL restores the
 * interpolant's analytic gradient (the default).  Built by the host from pp_recursive_gaussian_pass_f32. */
int pp_linear_set_moving_gradient(pp_ctx* ctx, const float* gradient, const int msize[3]);
/* Optional companion of the image above for the value + gradient kernel (pp_meansq_affine_f32 / pp_corr_moments_affine_f32 and
 * through them pp_linear_optimize_f32): the SAME gradient image packed with the moving image's intensity, [Z][Y][X][4] =
 * (gx, gy, gz, m) per voxel, 16-byte aligned (device, caller-owned, valid until replaced).  With ITK's jittered sample points
 * every sample has its own rows; one 16-byte element per corner is 8 gathers and ~0.25 KB of cache sectors a sample where the
 * planar images cost 32 and ~1 KB.  Results are bit-identical.  Call after pp_linear_set_moving_gradient (which clears it);
 * NULL removes it.  (No reference counterpart: a layout of this build.) */
int pp_linear_set_moving_gradient_packed(pp_ctx* ctx, const float* packed);

/* Mutual-information metrics (SetMetricAsMattesMutualInformation / SetMetricAsJointHistogramMutualInformation,
 * registration/linear.py:145-148) over the same sample lattice: pass 1 returns the joint intensity histogram of the valid
 * sample pairs (row = fixed bin; 64-bit fixed-point accumulation, independent of scheduling) and their count; the caller
 * turns it into PDFs, the value and a per-bin score table; pass 2 returns sum_s w_s g_s v_q / sum_s w_s g_s in the
 * d/dAm (9), d/dbm (3) layout of pp_meansq_affine_f32 with w_s = sum_k dkernel_k(s) table[f_bin(s)][k].
 * bin coordinate of an intensity = value / *_bin - *_norm_min.  PP_MI_MATTES: fixed nearest bin, moving cubic B-spline
 * over 4 bins, both clamped to [2, nbins - 3] (itk::MattesMutualInformationImageToImageMetricv4); PP_MI_JOINT: nearest
 * bins, the score differenced between the two neighbouring moving-bin centres. */
enum { PP_MI_MATTES = 0, PP_MI_JOINT = 1 };
typedef struct pp_mi_bins {
  int nbins;                 /* <= 64 */
  int kernel;                /* PP_MI_* */
  double f_bin, f_norm_min;  /* fixed:  bin width, normalised minimum */
  double m_bin, m_norm_min;  /* moving */
} pp_mi_bins;
int pp_mi_histogram_f32(pp_ctx* ctx, const float* fixed, const int fsize[3], const float* moving, const int msize[3],
                        const double Af[9], const double bf[3], const double Am[9], const double bm[3],
                        const int vsize[3], int stride, const uint8_t* fixed_mask, const uint8_t* moving_mask,
                        const pp_mi_bins* bins, double* hist, double* count);
int pp_mi_gradient_f32(pp_ctx* ctx, const float* fixed, const int fsize[3], const float* moving, const int msize[3],
                       const double Af[9], const double bf[3], const double Am[9], const double bm[3],
                       const int vsize[3], int stride, const uint8_t* fixed_mask, const uint8_t* moving_mask,
                       const pp_mi_bins* bins, const double* table, double* result);

/* One resolution level of linear_registration's optimisation (what registration.Execute does inside a level,
 * registration/linear.py:129-238): ITK v4 gradient descent (optionally with the golden-section line search) on
 * the mean-squares or correlation metric above, parameter scales from physical shift, learning rate estimated
 * once, convergence window 10 / 1e-6, best point kept.  Host logic in the library, metric on the GPU. */
enum { PP_MODEL_TRANSLATION = 0, PP_MODEL_VERSOR_RIGID = 1, PP_MODEL_SIMILARITY = 2, PP_MODEL_SCALE = 3,
       PP_MODEL_AFFINE = 4, PP_MODEL_EULER = 5, PP_MODEL_SCALE_VERSOR = 6,
       PP_MODEL_SCALE_SKEW_VERSOR = 7 };                 /* sitk parameter layouts; 3/6/7/3/12/6/9/15 parameters */
enum { PP_OPT_GD = 0, PP_OPT_GD_LINE_SEARCH = 1 };
enum { PP_LINREG_RETURN_BEST = 1 };
enum { PP_LINREG_STOP_ITERATIONS = 0, PP_LINREG_STOP_CONVERGED = 1, PP_LINREG_STOP_NO_OVERLAP = 2 };
typedef struct pp_linreg_level {
  int model;              /* PP_MODEL_*: q = A(params) (p - center) + center + t(params)            */
  int metric;             /* 0 mean squares, 1 correlation                                            */
  int optimizer;          /* PP_OPT_*                                                                 */
  int iterations;         /* numberOfIterations                                                       */
  int vsize[3];           /* virtual (shrunk fixed) domain                                            */
  int stride;             /* REGULAR sampling: every stride-th voxel of it                            */
  int speculation;        /* golden-section tree levels probed per launch, 1..4 (1 = sequential)      */
  int flags;              /* PP_LINREG_RETURN_BEST or 0 (ITK / SimpleITK default: the level's last point) */
  double v_i2p[9], v_origin[3]; /* virtual index -> physical: p = v_i2p idx + v_origin               */
  double f_p2i[9], f_origin[3]; /* fixed  physical -> index:  idx = f_p2i (p - f_origin)             */
  double m_p2i[9], m_origin[3]; /* moving physical -> index                                           */
  double init_matrix[9], init_offset[3]; /* the centring transform composed in front: q = M p + o    */
  double center[3];       /* fixed centre of rotation of the optimised transform                      */
  double v_min_spacing;   /* m_MaximumStepSizeInPhysicalUnits of the learning-rate estimate: the smallest virtual
                             spacing OF THE FIRST LEVEL (ITK assigns the default once per optimiser)     */
} pp_linreg_level;
typedef struct pp_linreg_stats {
  int iterations;         /* optimiser iterations taken            */
  int evaluations;        /* metric evaluations (probes included)  */
  int stop;               /* PP_LINREG_STOP_*                      */
  int reserved;
  double value;           /* optimiser's GetMetricValue(): the last evaluation (with RETURN_BEST: at the returned point) */
  double learning_rate;   /* last learning rate                    */
} pp_linreg_stats;
int pp_linear_num_parameters(int model);
/* params (host, in/out): pp_linear_num_parameters(model) doubles.  history (host, may be NULL): metric value
 * per iteration, up to history_capacity.  PP_ERR_NO_OVERLAP when no sample point is valid at the start. */
int pp_linear_optimize_f32(pp_ctx* ctx, const float* fixed, const int fsize[3], const float* moving,
                           const int msize[3], const uint8_t* fixed_mask, const uint8_t* moving_mask,
                           const pp_linreg_level* level, double* params, pp_linreg_stats* stats,
                           double* history, int history_capacity);

#ifdef __cplusplus
}
#endif
#endif /* PLATIPY_AMD_H */
