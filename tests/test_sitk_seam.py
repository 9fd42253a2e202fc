"""The SimpleITK-facing seams of the drop-in, executed against tests/sitk_double (a test double of the few sitk calls
involved -- SimpleITK itself cannot be installed here):

  * platipy_amd.image.from_sitk / to_sitk / as_image round trips (scalar, uint8, VectorFloat64);
  * seam 2 of INTEGRATION.md: HipDemonsFilter as the `registration_algorithm` of a multiscale loop written against the
    sitk API -- a restatement, in this file, of what the reference's multiscale_demons does with the filter
    (deformable.py:120-187): it receives sitk images, must return a VectorFloat64 sitk image the loop can pass to
    sitk.Resample(dvf_iter, tfm_total) and add to dvf_total;
  * the INTEGRATION.md section-2 snippet itself, extracted from the file and executed."""
import os
import re
import sys

import numpy as np
import pytest

from tests.helpers import phantom, random_dvf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def sitk(monkeypatch):
    monkeypatch.syspath_prepend(os.path.join(ROOT, "tests", "sitk_double"))
    sys.modules.pop("SimpleITK", None)
    import SimpleITK

    assert SimpleITK.__version__ == "test-double"
    yield SimpleITK
    sys.modules.pop("SimpleITK", None)


def _sitk_image(sitk, arr, spacing, origin, vec=False):
    im = sitk.GetImageFromArray(arr, isVector=vec)
    im.SetSpacing(spacing)
    im.SetOrigin(origin)
    return im


def test_sitk_round_trips(host_api, sitk):
    pa = host_api
    from platipy_amd.image import as_image, from_sitk, to_sitk

    sp, org = (0.9, 1.1, 2.5), (3.0, -4.0, 5.0)
    ct = phantom((6, 8, 10), seed=1)
    im = from_sitk(_sitk_image(sitk, ct, sp, org))
    assert isinstance(im, pa.Image) and im.GetSize() == (10, 8, 6) and im.spacing == sp and im.origin == org and not im.is_vector
    np.testing.assert_array_equal(im.numpy(), ct)
    back = to_sitk(im)
    assert back.GetPixelID() == sitk.sitkFloat32 and back.GetSpacing() == sp and back.GetOrigin() == org
    np.testing.assert_array_equal(sitk.GetArrayFromImage(back), ct)
    mask = (ct > 0).astype(np.uint8)
    m2 = to_sitk(from_sitk(_sitk_image(sitk, mask, sp, org)))
    assert m2.GetPixelID() == sitk.sitkUInt8
    np.testing.assert_array_equal(sitk.GetArrayFromImage(m2), mask)
    # a displacement field: sitk stores [Z, Y, X, 3] VectorFloat64, the product holds planar fp32 [3, Z, Y, X]
    dv = random_dvf((6, 8, 10), sp, seed=2).astype(np.float64)
    f = from_sitk(_sitk_image(sitk, np.ascontiguousarray(np.moveaxis(dv, 0, -1)), sp, org, vec=True))
    assert f.is_vector and tuple(f.tensor.shape) == (3, 6, 8, 10)
    np.testing.assert_allclose(f.numpy(), dv, rtol=1e-7)
    fb = to_sitk(f)
    assert fb.GetPixelID() == sitk.sitkVectorFloat64 and fb.GetNumberOfComponentsPerPixel() == 3
    np.testing.assert_allclose(np.moveaxis(sitk.GetArrayFromImage(fb), -1, 0), dv, rtol=1e-7)
    sitk.DisplacementFieldTransform(fb)                        # the constraint the reference's loop relies on (:139)
    assert as_image(_sitk_image(sitk, ct, sp, org)).GetSize() == (10, 8, 6)
    with pytest.raises(TypeError):
        as_image(ct)


def _shrink(sitk, image, factor, sigma):
    """The reference's smooth_and_resample for a scalar shrink factor (registration/utils.py:195-267), restated."""
    if sigma:
        var = (sigma ** 2,) * 3
        image = sitk.DiscreteGaussian(image, var, int(max(8 * v * s for s, v in zip(image.GetSpacing(), var))))
    size, spacing = image.GetSize(), image.GetSpacing()
    new_size = [int(n / float(factor) + 0.5) for n in size]
    new_spacing = [((n - 1) * s) / (m - 1) for n, s, m in zip(size, spacing, new_size)]
    return sitk.Resample(image, new_size, sitk.Transform(), sitk.sitkLinear, image.GetOrigin(), new_spacing, image.GetDirection(), 0.0,
                         image.GetPixelID())


def _multiscale_loop(sitk, registration_algorithm, fixed, moving, staging, iterations):
    """What the reference's multiscale_demons does around `registration_algorithm` (deformable.py:120-187), restated
    against the sitk API: the filter is handed sitk images and its result goes straight into sitk.Resample / `+`."""
    fixed_images = [_shrink(sitk, fixed, r, r) for r in staging]
    moving_images = [_shrink(sitk, moving, r, r) for r in staging]
    dvf_total = sitk.Image(fixed.GetWidth(), fixed.GetHeight(), fixed.GetDepth(), sitk.sitkVectorFloat64)
    dvf_total.CopyInformation(fixed)
    for f_image, m_image, iters in zip(fixed_images, moving_images, iterations):
        dvf_total = sitk.Resample(dvf_total, f_image)
        tfm_total = sitk.DisplacementFieldTransform(sitk.Cast(dvf_total, sitk.sitkVectorFloat64))
        m_image = sitk.Resample(m_image, tfm_total, sitk.sitkLinear)
        registration_algorithm.SetNumberOfIterations(iters)
        dvf_iter = registration_algorithm.Execute(f_image, m_image)
        dvf_total = dvf_total + sitk.Resample(dvf_iter, tfm_total)
        dvf_total = sitk.Cast(sitk.SmoothingRecursiveGaussian(dvf_total, registration_algorithm.GetStandardDeviations()),
                              sitk.sitkVectorFloat64)
    return sitk.Resample(dvf_total, fixed)


def _pair(shape, spacing, origin):
    from oracle import oracle as O

    fix = phantom(shape, seed=100)
    dv = random_dvf(shape, spacing, seed=101, max_mm=3.0)
    mov = O.warp_image(O.Vol(phantom(shape, seed=100, noise=0), spacing, origin), dv.astype(np.float64), edge_value=-1000.0).arr
    return fix, (mov + np.random.default_rng(102).normal(0, 5, size=shape)).astype(np.float32)


def test_hip_filter_inside_a_sitk_multiscale_loop(host_api, sitk):
    pa = host_api
    shape, sp, org = (16, 24, 40), (1.0, 1.1, 2.0), (10.0, -20.0, 5.0)
    fix, mov = _pair(shape, sp, org)
    fi, mi = _sitk_image(sitk, fix, sp, org), _sitk_image(sitk, mov, sp, org)
    flt = pa.registration.HipDemonsFilter()
    flt.SetSmoothUpdateField(True)
    flt.SetSmoothDisplacementField(True)
    flt.SetStandardDeviations([1.5 / s for s in sp])
    one = flt.Execute(fi, mi)                                         # sitk in -> sitk VectorFloat64 out, fixed grid
    assert type(one).__module__.startswith("SimpleITK") and one.GetPixelID() == sitk.sitkVectorFloat64
    assert one.GetSize() == fi.GetSize() and one.GetSpacing() == sp and one.GetOrigin() == org
    dvf = _multiscale_loop(sitk, flt, fi, mi, [4, 2, 1], [5, 5, 4])
    assert dvf.GetPixelID() == sitk.sitkVectorFloat64 and dvf.GetSize() == fi.GetSize()
    got = np.moveaxis(sitk.GetArrayFromImage(dvf), -1, 0)
    # the same registration through the product's own multiscale_demons on HBM-resident images
    want = pa.registration.multiscale_demons(registration_algorithm=flt, fixed_image=pa.image_from_array(fix, sp, org),
                                             moving_image=pa.image_from_array(mov, sp, org), resolution_staging=[4, 2, 1],
                                             smoothing_sigmas=[4, 2, 1], iteration_staging=[5, 5, 4]).numpy()
    err = np.abs(got - want)      # the loop's sitk.* steps run in the double's fp64 CPU arithmetic, the product's in fp32 HIP
    assert np.median(err) <= 5e-5 and np.quantile(err, 0.99) <= 1e-3 and np.sqrt((err ** 2).mean()) <= 2e-3
    assert np.abs(want).max() > 0.2


def test_integration_md_section_2_snippet_runs(host_api, sitk):
    """INTEGRATION.md section 2 says: keep multiscale_demons, swap only the filter.  The snippet is taken from the file."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2."):text.index("## 3.")]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    lines = [ln for ln in code.splitlines() if not ln.startswith("registration_method = sitk.")]   # "today" line
    scope = {}
    exec("\n".join(lines), scope)
    registration_method = scope["registration_method"]
    registration_method.SetSmoothUpdateField(True)                   # deformable.py:247-257, unchanged
    registration_method.SetSmoothDisplacementField(True)
    shape, sp, org = (12, 16, 24), (1.0, 1.0, 2.0), (0.0, 0.0, 0.0)
    registration_method.SetStandardDeviations([1.5 / s for s in sp])
    fix, mov = _pair(shape, sp, org)
    dvf = _multiscale_loop(sitk, registration_method, _sitk_image(sitk, fix, sp, org), _sitk_image(sitk, mov, sp, org), [2, 1], [4, 4])
    a = sitk.GetArrayFromImage(dvf)
    assert a.shape == shape + (3,) and a.dtype == np.float64 and np.isfinite(a).all() and np.abs(a).max() > 0.05


def _oriented_cases():
    ang = 0.3
    c, s = np.cos(ang), np.sin(ang)
    return {"x_flipped": (-1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0),
            "oblique": (c, 0.0, s, 0.0, 1.0, 0.0, -s, 0.0, c)}          # tilted about y, like a gantry-tilted series


@pytest.mark.parametrize("case", ["x_flipped", "oblique"])
def test_hip_filter_takes_oriented_sitk_images(host_api, sitk, case):
    """Seam 2 with non-identity direction cosines (missing 4): sitk's filter at deformable.py:149 takes any
    direction, so a maintainer who swaps only the filter must not meet an exception on an oblique CT.  The field that comes
    back is physical (LPS) on the fixed grid: R times the field of the identity-direction run on the same voxel arrays --
    checked against the ORACLE's Execute (not against the product's own identity run), the sitk loop around it included."""
    pa = host_api
    from oracle import oracle as O

    direction = _oriented_cases()[case]
    R = np.array(direction).reshape(3, 3)
    shape, sp, org = (16, 24, 40), (1.0, 1.1, 2.0), (10.0, -20.0, 5.0)
    fix, mov = _pair(shape, sp, org)
    fi, mi = _sitk_image(sitk, fix, sp, org), _sitk_image(sitk, mov, sp, org)
    fi.SetDirection(direction)
    mi.SetDirection(direction)
    flt = pa.registration.HipDemonsFilter()
    flt.SetSmoothUpdateField(True)
    flt.SetSmoothDisplacementField(True)
    flt.SetStandardDeviations([1.5 / s for s in sp])
    flt.SetNumberOfIterations(4)
    out = flt.Execute(fi, mi)
    assert type(out).__module__.startswith("SimpleITK") and out.GetPixelID() == sitk.sitkVectorFloat64
    assert out.GetSize() == fi.GetSize() and out.GetSpacing() == sp and out.GetOrigin() == org
    np.testing.assert_allclose(out.GetDirection(), direction, rtol=0, atol=0)
    got = np.moveaxis(sitk.GetArrayFromImage(out), -1, 0)
    # the oracle's Execute on the same voxel arrays in the index-aligned frame, rotated to physical components
    orc = O.DemonsFilter()
    orc.SetSmoothUpdateField(True)
    orc.SetSmoothDisplacementField(True)
    orc.SetStandardDeviations([1.5 / s for s in sp])
    orc.SetNumberOfIterations(4)
    local = orc.Execute(O.Vol(fix, sp, org), O.Vol(mov, sp, org)).arr
    want = np.einsum("rc,czyx->rzyx", R, local)
    err = np.abs(got - want)
    assert err.max() <= 2e-3 and np.sqrt((err ** 2).mean()) <= 5e-5, (err.max(), np.sqrt((err ** 2).mean()))   # test_demons_execute's bounds
    assert flt.GetElapsedIterations() == orc.GetElapsedIterations() and np.abs(want).max() > 0.1
    # different grids are still refused, as the reference's docstring demands (deformable.py:210-211)
    other = _sitk_image(sitk, mov, sp, org)
    with pytest.raises(ValueError):
        flt.Execute(fi, other)
    # the product's own multiscale_demons, called directly with oriented images, returns the physical field of the
    # identity-direction run as well
    kw = dict(resolution_staging=[2, 1], smoothing_sigmas=[2, 1], iteration_staging=[3, 3])
    d0 = pa.registration.multiscale_demons(registration_algorithm=flt, fixed_image=pa.image_from_array(fix, sp, org),
                                           moving_image=pa.image_from_array(mov, sp, org), **kw)
    d1 = pa.registration.multiscale_demons(registration_algorithm=flt, fixed_image=pa.image_from_array(fix, sp, org, direction),
                                           moving_image=pa.image_from_array(mov, sp, org, direction), **kw)
    assert d1.direction == direction
    np.testing.assert_allclose(d1.numpy(), np.einsum("rc,czyx->rzyx", R, d0.numpy().astype(np.float64)), rtol=0, atol=1e-5)
