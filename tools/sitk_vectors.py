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
    version = sitk.Version.VersionString() if hasattr(sitk, "Version") else getattr(sitk, "__version__", "unknown")
    out["meta_generator"] = np.array(generator)
    out["meta_sitk_version"] = np.array(str(version))
    out["meta_grid"] = np.array(list(SHAPE) + list(SPACING) + list(ORIGIN), dtype=np.float64)
    np.savez_compressed(path, **out)
    return out


def is_reference(vectors):
    """True for a file written by the real SimpleITK; False for the test double's (plumbing only)."""
    return str(vectors["meta_generator"]) == "SimpleITK" and "test-double" not in str(vectors["meta_sitk_version"])
