#!/usr/bin/env python
"""Reference vectors for the hot path, generated with the PUBLIC SimpleITK API ("next" item 2).

`emit(sitk, path)` runs, on tests/golden/make_golden.py's seeded inputs, exactly the SimpleITK calls the reference makes
on this path and stores their outputs:

  execute_1it / execute_4it / execute_halt   sitk.FastSymmetricForcesDemonsRegistrationFilter configured as
                                             registration/deformable.py:244-257, Execute as :149 -- 1 iteration, 4 iterations
                                             (default MaximumRMSError), and 6 iterations with a MaximumRMSError that the
                                             RMS change falls below after a few iterations (the Halt() rule)
  recursive_gaussian                         sitk.SmoothingRecursiveGaussian(dvf, sigma)             deformable.py:157-158
  discrete_gaussian_var4 / _var1             sitk.DiscreteGaussian(image, variance)                  registration/utils.py:226, fusion.py:168,279
  resample_linear / resample_nearest         sitk.Resample(image, image, DisplacementFieldTransform, interp, default)
                                                                                                     registration/utils.py:176-190
  distance_map_signed / label_contour        sitk.SignedMaurerDistanceMap / sitk.LabelContour        label/projection.py:80-90

Round 4 -- one key (or key group) per remaining SURVEY 8 row, so that the day the command runs every (a)/(f) row is pinned:

  pyramid_level                              one level of smooth_and_resample: sitk.DiscreteGaussian(image, sigma^2 (x3),
                                             maximumKernelWidth = int(max(8 sigma^2 spacing))) then the corner-aligned
                                             sitk.Resample(image, new_size, Transform(), sitkLinear, origin, new_spacing,
                                             direction, 0.0, pixel id)                               registration/utils.py:216-267
  weight_local / weight_block                Pow(DiscreteGaussian(SquaredDifference, 4) + 1e-5, -1);
                                             factor * Pow(BoxMean(SquaredDifference, 5), -1) ** |gain / 2|  label/fusion.py:148-190
  fused_probability / fused_mask             three atlases (the mask shifted, the moving image shifted, local weights): weighted
                                             vote / guarded weight sum, DiscreteGaussian(1), RescaleIntensity(0, 1),
                                             Threshold(1e-4); then / max, BinaryThreshold(0.5), BinaryFillhole,
                                             ConnectedComponent, the largest component              label/fusion.py:263-288, 305-328
  dilate_ball_221 / close_ball_210           sitk.BinaryDilate / sitk.BinaryMorphologicalClosing, ball element
                                                                                                     registration/utils.py:328-329, cardiac/run.py:1121-1127
  fillhole_component                         sitk.BinaryFillhole + sitk.ConnectedComponent of a mask with a cavity and a satellite
  linear_metric_value, _masked_value,        sitk.ImageRegistrationMethod (mean squares, every voxel, linear interpolation):
  _masked_fd_gradient                        MetricEvaluate at a fixed affine map, the same under a fixed-image mask, and central
                                             differences of the masked value along the 12 AffineTransform parameters
                                                                                                     registration/linear.py:133-163
  linear_similarity_corners                  the 8 corners of the fixed image mapped by the result of the reference's
                                             linear_registration call (similarity, mean squares, gradient descent, shrink
                                             [2, 1], sigmas [1, 0], sampling 1.0, 20 iterations)     registration/linear.py:125-238
A call the installed SimpleITK (or the test double) cannot make is recorded in meta_missing instead of failing the file.

The file this writes where SimpleITK exists (tools/compare_with_sitk.py --emit tests/golden/sitk_<version>.npz) is DATA:
inputs' seeds and SimpleITK's outputs.  tests/test_golden.py picks up every tests/golden/sitk_*.npz and holds BOTH the
oracle and the product to it; committing one such file turns "parity unpinned" into a reference-pinned test without any
code change.  The module takes the `sitk` module as an argument so that the CPU suite can run the same code against
tests/sitk_double (plumbing check only: that double is backed by the oracle and proves nothing about ITK)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.golden.make_golden import ORIGIN, SHAPE, SPACING, inputs  # noqa: E402

SIGMA = [1.5 / s for s in SPACING]      # deformable.py:253-257: regularisation_kernel_mm / spacing
HALT_ITERATIONS = 6
HALT_MAX_RMS = 0.3                      # the RMS change of this pair (0.62, 0.42, 0.31, 0.284, ...) falls below it after 4 iterations


def _image(sitk, arr, vector=False):
    a = np.ascontiguousarray(np.moveaxis(arr, 0, -1)).astype(np.float64) if vector else np.ascontiguousarray(arr)
    img = sitk.GetImageFromArray(a, isVector=vector)
    img.SetSpacing(SPACING)
    img.SetOrigin(ORIGIN)
    return img


def _demons(sitk, n, max_rms=None):
    flt = sitk.FastSymmetricForcesDemonsRegistrationFilter()
    flt.SetSmoothUpdateField(True)               # deformable.py:248
    flt.SetSmoothDisplacementField(True)         # :249
    flt.SetStandardDeviations(SIGMA)             # :253-257
    flt.SetNumberOfIterations(n)                 # :144
    if max_rms is not None:
        flt.SetMaximumRMSError(max_rms)
    return flt


# ---- round 4: the remaining SURVEY 8 rows -------------------------------------------------------------------------
PYRAMID_SIGMA_MM, PYRAMID_SHRINK = 2.0, 2                     # one level of the pyramid on the small grid
BLOCK_PARAMS = {"factor": 1e12, "gain": 6, "blockSize": 5}    # fusion.py's default vote_params for "block" (reference :60-70)
ATLAS_SHIFTS = ((0, 0, 0), (0, 1, -1), (1, -1, 0))            # (z, y, x) rolls that make three "atlases" of the seeded pair
AFFINE_A = ((1.02, 0.03, -0.01), (-0.02, 0.97, 0.015), (0.01, -0.005, 1.03))   # the fixed affine map of the metric check
AFFINE_T = (0.6, -0.4, 0.3)
FD_STEP_MATRIX, FD_STEP_MM = 1e-3, 1e-2
LINEAR_KW = dict(reg_method="similarity", metric="mean_squares", optimiser="gradient_descent", shrink_factors=[2, 1], smooth_sigmas=[1, 0],
                 sampling_rate=1.0, number_of_iterations=20)


# Round 6: optimiser TRAJECTORIES of sitk.ImageRegistrationMethod as the reference configures it (registration/linear.py:133-238),
# per case: reg_method, optimiser; two levels without smoothing, REGULAR sampling at 0.5 with seed 42, 12 iterations a level.
TRAJECTORY_KW = dict(metric="mean_squares", shrink_factors=[2, 1], smooth_sigmas=[0, 0], sampling_rate=0.5, number_of_iterations=12)
TRAJECTORY_CASES = (("rigid", "gradient_descent_line_search"), ("affine", "gradient_descent_line_search"),
                    ("similarity", "gradient_descent"))


def cavity_mask(mask):
    """the seeded mask with a closed cavity inside it and a small satellite beside it (fill-hole / largest-component input)"""
    m = mask.copy()
    m[5, 7, 9] = 0
    m[1:3, 1:3, 1:3] = 1
    return m


def atlas_inputs(fixed, moving, mask):
    """(moving image, label) of the three synthetic atlases: whole-voxel rolls of the seeded moving image and mask"""
    return [(np.roll(moving, sh, axis=(0, 1, 2)).copy(), np.roll(mask, sh, axis=(0, 1, 2)).copy()) for sh in ATLAS_SHIFTS]


def image_centre():
    return tuple(ORIGIN[k] + 0.5 * (SHAPE[2 - k] - 1) * SPACING[k] for k in range(3))


def _local_weight(sitk, T, M):
    sq = sitk.Cast(sitk.SquaredDifference(T, M), sitk.sitkFloat32)
    return sitk.Cast(sitk.Pow(sitk.DiscreteGaussian(sq, 2.0 * 2.0) + 1e-5, -1.0), sitk.sitkFloat32)


def _metric(sitk, F, M, A, t, fixed_mask=None):
    R = sitk.ImageRegistrationMethod()
    R.SetMetricAsMeanSquares()
    if fixed_mask is not None:
        R.SetMetricFixedMask(fixed_mask)          # registration/linear.py:162-163
    R.SetMetricSamplingStrategy(R.NONE)
    R.SetInterpolator(sitk.sitkLinear)
    T = sitk.AffineTransform(3)
    T.SetCenter(image_centre())
    T.SetMatrix([float(v) for v in np.asarray(A).ravel()])
    T.SetTranslation([float(v) for v in t])
    R.SetInitialTransform(T, inPlace=False)
    return float(R.MetricEvaluate(F, M))


def emit_round4(sitk, out, fixed, moving, field, mask):
    """Adds the round-4 keys to `out`; -> names of the groups this sitk could not produce."""
    missing = []
    F, M, K = _image(sitk, fixed), _image(sitk, moving), _image(sitk, mask)

    def group(name, fn):
        try:
            fn()
        except (AttributeError, NotImplementedError, TypeError) as e:
            missing.append(f"{name}: {type(e).__name__}: {e}")

    def pyramid():
        var = [PYRAMID_SIGMA_MM ** 2] * 3
        width = int(max(8 * v * sp for v, sp in zip(var, SPACING)))
        img = sitk.DiscreteGaussian(F, var, width)
        size, spacing = img.GetSize(), img.GetSpacing()
        new_size = [int(sz / float(PYRAMID_SHRINK) + 0.5) for sz in size]
        new_spacing = [((so - 1) * sp) / (sn - 1) for so, sp, sn in zip(size, spacing, new_size)]
        r = sitk.Resample(img, new_size, sitk.Transform(), sitk.sitkLinear, img.GetOrigin(), new_spacing, img.GetDirection(), 0.0, img.GetPixelID())
        out["pyramid_level"] = sitk.GetArrayFromImage(r).astype(np.float32)
        out["pyramid_level_spacing"] = np.array(r.GetSpacing(), dtype=np.float64)

    def weights():
        out["weight_local"] = sitk.GetArrayFromImage(_local_weight(sitk, F, M)).astype(np.float32)
        sq = sitk.Cast(sitk.SquaredDifference(F, M), sitk.sitkFloat32)
        raw = sitk.BoxMean(sq, (BLOCK_PARAMS["blockSize"],) * 3)
        w = BLOCK_PARAMS["factor"] * sitk.Pow(raw, -1.0) ** abs(BLOCK_PARAMS["gain"] / 2.0)
        out["weight_block"] = sitk.GetArrayFromImage(sitk.Cast(w, sitk.sitkFloat32)).astype(np.float32)

    def fusion():
        atl = [(_image(sitk, m), _image(sitk, l)) for m, l in atlas_inputs(fixed, moving, mask)]
        ws = [_local_weight(sitk, F, m) for m, _ in atl]
        wsum = ws[0] + ws[1] + ws[2]
        wsum = sitk.Mask(wsum, wsum == 0, maskingValue=1, outsideValue=1)
        wl = [w * sitk.Cast(l, sitk.sitkFloat32) for w, (_, l) in zip(ws, atl)]
        p = (wl[0] + wl[1] + wl[2]) / wsum
        p = sitk.DiscreteGaussian(p, 1.0 * 1.0)
        p = sitk.RescaleIntensity(p, 0, 1)
        p = sitk.Threshold(p, lower=1e-4, upper=1, outsideValue=0.0)
        out["fused_probability"] = sitk.GetArrayFromImage(p).astype(np.float32)
        q = p / float(sitk.GetArrayFromImage(p).max())
        b = sitk.BinaryFillhole(sitk.BinaryThreshold(q, lowerThreshold=0.5))
        lab = sitk.ConnectedComponent(b)
        arr = sitk.GetArrayFromImage(lab)
        counts = np.bincount(arr.ravel())
        counts[0] = 0
        out["fused_mask"] = (arr == int(np.argmax(counts))).astype(np.uint8)

    def morphology():
        out["dilate_ball_221"] = sitk.GetArrayFromImage(sitk.BinaryDilate(K, (2, 2, 1), sitk.sitkBall)).astype(np.uint8)
        from tests.golden.make_golden import NOTCHED

        out["close_ball_210"] = sitk.GetArrayFromImage(sitk.BinaryMorphologicalClosing(_image(sitk, NOTCHED(mask)), (2, 1, 0), sitk.sitkBall)).astype(np.uint8)

    def components():
        C = _image(sitk, cavity_mask(mask))
        filled = sitk.BinaryFillhole(C)
        out["fillhole"] = sitk.GetArrayFromImage(filled).astype(np.uint8)
        out["fillhole_component"] = sitk.GetArrayFromImage(sitk.ConnectedComponent(filled)).astype(np.int32)

    def metric():
        A, t = np.array(AFFINE_A, dtype=np.float64), np.array(AFFINE_T, dtype=np.float64)
        Ff, Mf = sitk.Cast(F, sitk.sitkFloat32), sitk.Cast(M, sitk.sitkFloat32)
        out["linear_metric_value"] = np.array(_metric(sitk, Ff, Mf, A, t))
        # The derivative is not exposed by SimpleITK: central differences of MetricEvaluate, taken under a fixed-image mask
        # (the seeded structure: SetMetricFixedMask, linear.py:162-163) so that no sample enters or leaves the moving image
        # between the two evaluations -- on the whole grid the set of valid samples changes with the parameters and the mean
        # jumps with it.
        out["linear_metric_masked_value"] = np.array(_metric(sitk, Ff, Mf, A, t, K))
        g = np.zeros(12)
        for k in range(12):
            dA, dt = np.zeros(9), np.zeros(3)
            h = FD_STEP_MATRIX if k < 9 else FD_STEP_MM
            (dA if k < 9 else dt)[k if k < 9 else k - 9] = h
            g[k] = (_metric(sitk, Ff, Mf, A + dA.reshape(3, 3), t + dt, K) - _metric(sitk, Ff, Mf, A - dA.reshape(3, 3), t - dt, K)) / (2 * h)
        out["linear_metric_masked_fd_gradient"] = g

    def similarity():
        # registration/linear.py:125-238, the sitk calls of the reference's function for LINEAR_KW
        Ff, Mf = sitk.Cast(F, sitk.sitkFloat32), sitk.Cast(M, sitk.sitkFloat32)
        init = sitk.CenteredTransformInitializer(Ff, Mf, sitk.Euler3DTransform(), False)
        R = sitk.ImageRegistrationMethod()
        R.SetShrinkFactorsPerLevel(LINEAR_KW["shrink_factors"])
        R.SetSmoothingSigmasPerLevel(LINEAR_KW["smooth_sigmas"])
        R.SmoothingSigmasAreSpecifiedInPhysicalUnitsOn()
        R.SetMovingInitialTransform(init)
        R.SetMetricAsMeanSquares()
        R.SetInterpolator(sitk.sitkLinear)
        R.SetMetricSamplingPercentage(LINEAR_KW["sampling_rate"], seed=42)
        R.SetMetricSamplingStrategy(sitk.ImageRegistrationMethod.REGULAR)
        R.SetOptimizerScalesFromPhysicalShift()
        R.SetInitialTransform(sitk.Similarity3DTransform())
        R.SetOptimizerAsGradientDescent(learningRate=1.0, numberOfIterations=LINEAR_KW["number_of_iterations"])
        tfm = sitk.CompositeTransform([init, R.Execute(fixed=Ff, moving=Mf)])
        corners = [[ORIGIN[k] + (SHAPE[2 - k] - 1) * SPACING[k] * ((c >> k) & 1) for k in range(3)] for c in range(8)]
        out["linear_similarity_corners"] = np.array([tfm.TransformPoint(c) for c in corners], dtype=np.float64)
        out["linear_similarity_metric"] = np.array(float(R.GetMetricValue()))

    def trajectories():
        # the same sitk calls with an iteration observer: the metric value ITK reports at every iteration (GetMetricValue() = the
        # evaluation at the head of that iteration), the level each belongs to, the optimised transform's final parameters and
        # where the composite sends the fixed image's corners -- what tests/test_linear_oracle.py compares product and oracle on
        Ff, Mf = sitk.Cast(F, sitk.sitkFloat32), sitk.Cast(M, sitk.sitkFloat32)
        corners = [[ORIGIN[k] + (SHAPE[2 - k] - 1) * SPACING[k] * ((c >> k) & 1) for k in range(3)] for c in range(8)]
        for method, optimiser in TRAJECTORY_CASES:
            init = sitk.CenteredTransformInitializer(Ff, Mf, sitk.Euler3DTransform(), False)
            R = sitk.ImageRegistrationMethod()
            R.SetShrinkFactorsPerLevel(TRAJECTORY_KW["shrink_factors"])
            R.SetSmoothingSigmasPerLevel(TRAJECTORY_KW["smooth_sigmas"])
            R.SmoothingSigmasAreSpecifiedInPhysicalUnitsOn()
            R.SetMovingInitialTransform(init)
            R.SetMetricAsMeanSquares()
            R.SetInterpolator(sitk.sitkLinear)
            R.SetMetricSamplingPercentage(TRAJECTORY_KW["sampling_rate"], seed=42)
            R.SetMetricSamplingStrategy(sitk.ImageRegistrationMethod.REGULAR)
            R.SetOptimizerScalesFromPhysicalShift()
            R.SetInitialTransform({"rigid": sitk.VersorRigid3DTransform, "similarity": sitk.Similarity3DTransform,
                                   "affine": lambda: sitk.AffineTransform(3)}[method]())
            if optimiser == "gradient_descent_line_search":
                R.SetOptimizerAsGradientDescentLineSearch(learningRate=1.0, numberOfIterations=TRAJECTORY_KW["number_of_iterations"])
            else:
                R.SetOptimizerAsGradientDescent(learningRate=1.0, numberOfIterations=TRAJECTORY_KW["number_of_iterations"])
            seen = []      # (level, optimiser iteration, metric value); an iteration may fire its event more than once

            def observe(R=R, seen=seen):
                seen.append((R.GetCurrentLevel(), R.GetOptimizerIteration(), R.GetMetricValue()))

            R.AddCommand(sitk.sitkIterationEvent, observe)
            result = R.Execute(fixed=Ff, moving=Mf)
            tfm = sitk.CompositeTransform([init, result])
            key = f"linear_trajectory_{method}_{optimiser}"
            first = {}
            for level, it, value in seen:
                first.setdefault((int(level), int(it)), float(value))
            out[key + "_values"] = np.array([[lv, it, v] for (lv, it), v in sorted(first.items())], dtype=np.float64)
            out[key + "_parameters"] = np.array(result.GetParameters(), dtype=np.float64)
            out[key + "_corners"] = np.array([tfm.TransformPoint(c) for c in corners], dtype=np.float64)
            out[key + "_stop"] = np.array(str(R.GetOptimizerStopConditionDescription()))

    for name, fn in (("pyramid_level", pyramid), ("weight maps", weights), ("fusion chain", fusion), ("ball morphology", morphology),
                     ("fill-hole / components", components), ("linear metric", metric), ("linear similarity registration", similarity),
                     ("linear optimiser trajectories", trajectories)):
        group(name, fn)
    return missing


def emit(sitk, path, generator="SimpleITK"):
    fixed, moving, field, mask = inputs()
    F, M, K = _image(sitk, fixed), _image(sitk, moving), _image(sitk, mask)
    D = _image(sitk, field, vector=True)
    out = {"fixed": fixed, "moving": moving, "field": field, "mask": mask}
    for key, n, max_rms in (("execute_1it", 1, None), ("execute_4it", 4, None), ("execute_halt", HALT_ITERATIONS, HALT_MAX_RMS)):
        flt = _demons(sitk, n, max_rms)
        dvf = sitk.GetArrayFromImage(flt.Execute(F, M))
        out[key] = np.ascontiguousarray(np.moveaxis(dvf, -1, 0)).astype(np.float64)
        out[key + "_stats"] = np.array([flt.GetElapsedIterations(), flt.GetMetric(), flt.GetRMSChange(), n,
                                        np.nan if max_rms is None else max_rms], dtype=np.float64)
    rg = sitk.GetArrayFromImage(sitk.SmoothingRecursiveGaussian(D, SIGMA))
    out["recursive_gaussian"] = np.ascontiguousarray(np.moveaxis(rg, -1, 0)).astype(np.float64)
    out["discrete_gaussian_var4"] = sitk.GetArrayFromImage(sitk.DiscreteGaussian(F, 4.0)).astype(np.float32)
    out["discrete_gaussian_var1"] = sitk.GetArrayFromImage(sitk.DiscreteGaussian(F, 1.0)).astype(np.float32)
    tfm = sitk.DisplacementFieldTransform(_image(sitk, field, vector=True))      # (the transform takes ownership of its image)
    out["resample_linear"] = sitk.GetArrayFromImage(sitk.Resample(M, M, tfm, sitk.sitkLinear, -1000.0)).astype(np.float32)
    tfm = sitk.DisplacementFieldTransform(_image(sitk, field, vector=True))
    out["resample_nearest"] = sitk.GetArrayFromImage(sitk.Resample(K, K, tfm, sitk.sitkNearestNeighbor, 0)).astype(np.uint8)
    out["distance_map_signed"] = sitk.GetArrayFromImage(
        sitk.SignedMaurerDistanceMap(K, insideIsPositive=False, squaredDistance=False, useImageSpacing=True)).astype(np.float32)
    out["label_contour"] = sitk.GetArrayFromImage(sitk.LabelContour(K)).astype(np.uint8)
    out["meta_missing"] = np.array(emit_round4(sitk, out, fixed, moving, field, mask))
    version = sitk.Version.VersionString() if hasattr(sitk, "Version") else getattr(sitk, "__version__", "unknown")
    out["meta_generator"] = np.array(generator)
    out["meta_sitk_version"] = np.array(str(version))
    out["meta_grid"] = np.array(list(SHAPE) + list(SPACING) + list(ORIGIN), dtype=np.float64)
    np.savez_compressed(path, **out)
    return out


def is_reference(vectors):
    """True for a file written by the real SimpleITK; False for the test double's (plumbing only)."""
    return str(vectors["meta_generator"]) == "SimpleITK" and "test-double" not in str(vectors["meta_sitk_version"])
