#!/usr/bin/env python
"""Pin the oracle (and, on a GPU box, the product) against the REAL reference arithmetic: SimpleITK.

The build image has no SimpleITK, so the repo's oracle is "parity unpinned" (DESIGN.md 3).  Run this
wherever `import SimpleITK` works: it regenerates per-stage vectors with the public SimpleITK API, configured
exactly as platipy does (registration/deformable.py:244-257,149; registration/utils.py:216-267;
label/fusion.py:163-169), and reports max / RMS differences against oracle/ (and platipy_amd if a GPU is
present).  It needs none of platipy's own files.  If SimpleITK is missing it says so and exits non-zero --
it never passes silently.

  python tools/compare_with_sitk.py                                   # report differences, stage by stage
  python tools/compare_with_sitk.py --emit tests/golden/sitk_2.3.1.npz  # write the reference vectors (tools/sitk_vectors.py);
                                                                        # tests/test_golden.py picks the file up
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

try:
    import SimpleITK as sitk
except ImportError:
    print("oracle unavailable: SimpleITK is not importable here; nothing was compared")
    sys.exit(2)

from oracle import oracle as O  # noqa: E402
from tests.helpers import phantom, random_dvf  # noqa: E402


def to_sitk(arr, spacing, origin, vector=False):
    img = sitk.GetImageFromArray(np.moveaxis(arr, 0, -1).astype(np.float64) if vector else arr, isVector=vector)
    img.SetSpacing(spacing)
    img.SetOrigin(origin)
    return img


def report(name, got, want):
    d = np.abs(np.asarray(got, dtype=np.float64) - np.asarray(want, dtype=np.float64))
    print(f"{name:42s} max {d.max():.3e}  rms {np.sqrt((d ** 2).mean()):.3e}")
    return d.max()


def main():
    shape, spacing, origin = (40, 56, 72), (0.98, 0.98, 2.5), (-120.0, -80.0, 30.0)
    fix = phantom(shape, seed=1)
    dv = random_dvf(shape, spacing, seed=2, max_mm=4.0)
    mov = O.warp_image(O.Vol(phantom(shape, seed=1, noise=0), spacing, origin), dv.astype(np.float64), edge_value=-1000.0).arr
    mov = (mov + np.random.default_rng(3).normal(0, 5, size=shape)).astype(np.float32)
    F, M = to_sitk(fix, spacing, origin), to_sitk(mov, spacing, origin)
    vf, vm = O.Vol(fix, spacing, origin), O.Vol(mov, spacing, origin)

    # 1. Gaussian operator / DiscreteGaussian (registration/utils.py:226, label/fusion.py:168)
    for var in (1.0, 4.0, 64.0):
        want = sitk.GetArrayFromImage(sitk.DiscreteGaussian(F, var, 64))
        report(f"DiscreteGaussian var={var}", O.discrete_gaussian(vf, var, 64).arr, want)
    # 2. one demons iteration and a 10-iteration Execute (deformable.py:244-257,149)
    for n in (1, 10):
        flt = sitk.FastSymmetricForcesDemonsRegistrationFilter()
        flt.SetSmoothUpdateField(True)
        flt.SetSmoothDisplacementField(True)
        flt.SetStandardDeviations([1.5 / s for s in spacing])
        flt.SetNumberOfIterations(n)
        want = np.moveaxis(sitk.GetArrayFromImage(flt.Execute(F, M)), -1, 0)
        o = O.DemonsFilter()
        o.SetSmoothUpdateField(True)
        o.SetStandardDeviations([1.5 / s for s in spacing])
        o.SetNumberOfIterations(n)
        got = o.Execute(vf, vm).arr
        report(f"demons Execute, {n} iteration(s)", got, want)
        print(f"    elapsed {o.GetElapsedIterations()} vs {flt.GetElapsedIterations()}  metric {o.GetMetric():.6f} vs {flt.GetMetric():.6f}"
              f"  rms {o.GetRMSChange():.6f} vs {flt.GetRMSChange():.6f}")
    # 3. SmoothingRecursiveGaussian on a vector field (deformable.py:157-158)
    sig = [1.5 / s for s in spacing]
    want = np.moveaxis(sitk.GetArrayFromImage(sitk.SmoothingRecursiveGaussian(to_sitk(dv, spacing, origin, True), sig)), -1, 0)
    report("SmoothingRecursiveGaussian(vector)", O.recursive_gaussian_vec(O.Vol(dv.astype(np.float64), spacing, origin), sig).arr, want)
    # 4. Resample through a DisplacementFieldTransform, linear and nearest (registration/utils.py:176-190)
    tfm = sitk.DisplacementFieldTransform(to_sitk(dv, spacing, origin, True))
    for interp, name in ((sitk.sitkLinear, "linear"), (sitk.sitkNearestNeighbor, "nearest")):
        want = sitk.GetArrayFromImage(sitk.Resample(M, M, tfm, interp, -1000.0))
        got = O.resample(vm, vm, field_vol=O.Vol(dv.astype(np.float64), spacing, origin),
                         interp=O.INTERP_LINEAR if interp == sitk.sitkLinear else O.INTERP_NEAREST, default_value=-1000.0).arr
        report(f"Resample through DVF ({name})", got, want)
    # 5. smooth_and_resample pyramid level (registration/utils.py:195-267)
    want = sitk.GetArrayFromImage(sitk.Resample(sitk.DiscreteGaussian(F, 16.0, int(8 * 16 * 2.5)), [18, 14, 10], sitk.Transform(),
                                                sitk.sitkLinear, F.GetOrigin(),
                                                [(72 - 1) * spacing[0] / 17, (56 - 1) * spacing[1] / 13, (40 - 1) * spacing[2] / 9],
                                                F.GetDirection(), 0.0, F.GetPixelID()))
    report("smooth_and_resample(shrink 4, sigma 4)", O.smooth_and_resample(vf, shrink_factor=4, smoothing_sigma=4).arr, want)
    # 6. local weight map (label/fusion.py:163-169)
    sq = sitk.Cast(sitk.SquaredDifference(F, M), sitk.sitkFloat32)
    want = sitk.GetArrayFromImage(sitk.Cast(sitk.Pow(sitk.DiscreteGaussian(sq, 4.0) + 1e-5, -1.0), sitk.sitkFloat32))
    got = O.compute_weight_map(vf, vm, "local").arr
    print(f"{'compute_weight_map(local), relative':42s} max {np.abs(got / want - 1).max():.3e}")

    # 7. binary morphology with the ball kernel (registration/utils.py:328-329; multiatlas/run.py:421-423)
    mask = (phantom(shape, seed=5, noise=0) > -300).astype(np.uint8)
    mask[:, :3, :] = 0
    Mk = to_sitk(mask, spacing, origin)
    mv = O.Vol(mask, spacing, origin)
    for radius in ((1, 1, 1), (2, 2, 0), (3, 2, 1)):
        report(f"BinaryDilate ball {radius}", O.binary_dilate_ball(mv, radius).arr, sitk.GetArrayFromImage(sitk.BinaryDilate(Mk, radius)))
        report(f"BinaryErode ball {radius}", O.binary_erode_ball(mv, radius).arr, sitk.GetArrayFromImage(sitk.BinaryErode(Mk, radius)))
        report(f"BinaryMorphologicalClosing ball {radius}", O.binary_closing_ball(mv, radius).arr,
               sitk.GetArrayFromImage(sitk.BinaryMorphologicalClosing(Mk, radius)))
    # 8. distance map, contour, fill-hole + largest component (label/projection.py:80-90, label/fusion.py:305-328)
    want = sitk.GetArrayFromImage(sitk.SignedMaurerDistanceMap(Mk, insideIsPositive=True, squaredDistance=False, useImageSpacing=True))
    report("SignedMaurerDistanceMap (inside positive)", O.maurer_distance_map(mv, signed=True, inside_positive=True).arr, want)
    report("LabelContour", O.label_contour(mv).arr, sitk.GetArrayFromImage(sitk.LabelContour(Mk)))
    prob = sitk.Cast(sitk.DiscreteGaussian(sitk.Cast(Mk, sitk.sitkFloat32), 2.0), sitk.sitkFloat32)
    pm = prob / float(sitk.GetArrayViewFromImage(prob).max())
    b = sitk.BinaryFillhole(sitk.BinaryThreshold(pm, lowerThreshold=0.5, upperThreshold=1.0))
    cc = sitk.ConnectedComponent(b)
    st = sitk.LabelShapeStatisticsImageFilter()
    st.Execute(cc)
    best = max(st.GetLabels(), key=st.GetNumberOfPixels) if st.GetLabels() else 0
    want = sitk.GetArrayFromImage(sitk.Cast(cc == best, sitk.sitkUInt8)) if best else sitk.GetArrayFromImage(b)
    report("process_probability_image(0.5)", O.process_probability_image(O.Vol(sitk.GetArrayFromImage(prob), spacing, origin), 0.5).arr, want)
    # 9. the mean-squares metric at a given affine map, and a whole linear_registration (registration/linear.py:129-238)
    reg = sitk.ImageRegistrationMethod()
    reg.SetMetricAsMeanSquares()
    reg.SetMetricSamplingStrategy(reg.NONE)
    reg.SetInterpolator(sitk.sitkLinear)
    tfm0 = sitk.AffineTransform(3)
    tfm0.SetMatrix([1.01, 0.02, 0.0, -0.015, 0.99, 0.01, 0.0, 0.005, 1.0])
    tfm0.SetTranslation((1.5, -2.0, 0.7))
    reg.SetInitialTransform(tfm0, inPlace=False)
    from oracle import linear_oracle

    A = np.array(tfm0.GetMatrix()).reshape(3, 3)
    t = np.array(tfm0.GetTranslation())
    i2p = np.diag(spacing)
    p2i = np.linalg.inv(i2p)
    o = np.array(origin)
    Am, bm = p2i @ A @ i2p, p2i @ (A @ o + t - o)
    r = linear_oracle.meansq_affine(fix, mov, np.eye(3), np.zeros(3), Am, bm, shape[::-1], 1)
    print(f"{'MeanSquares metric at a fixed affine map':42s} oracle {r[0] / r[1]:.6f}  sitk {reg.MetricEvaluate(F, M):.6f}")
    print("linear_registration end-to-end: run platipy.imaging.registration.linear.linear_registration and "
          "platipy_amd.registration.linear_registration on the same pair on a GPU box and compare the corner displacements "
          "(tests/test_linear.py::test_linear_registration_recovers_known_transform holds the build to < 1 mm of the known map)")

    try:
        import torch

        if torch.cuda.is_available():
            import platipy_amd as pa

            _, _, dvf = pa.registration.fast_symmetric_forces_demons_registration(pa.image_from_array(fix, spacing, origin),
                                                                                  pa.image_from_array(mov, spacing, origin))
            import platipy.imaging.registration.deformable as ref  # noqa: F401  (only if platipy itself is installed)
    except Exception as e:  # pragma: no cover
        print("GPU / platipy comparison skipped:", e)
    return 0


if __name__ == "__main__":
    if "--emit" in sys.argv:      # tools/compare_with_sitk.py --emit tests/golden/sitk_<version>.npz : the one-command hand-off
        from tools import sitk_vectors

        dest = sys.argv[sys.argv.index("--emit") + 1]
        sitk_vectors.emit(sitk, dest)
        print(f"wrote {dest} (SimpleITK {sitk.Version.VersionString()}); commit it: tests/test_golden.py then holds the oracle "
              "and the product to it")
        sys.exit(0)
    sys.exit(main())
