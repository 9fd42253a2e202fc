"""TEST DOUBLE -- a minimal stand-in for the SimpleITK Python API, for the CPU/GPU test-suite only.

SimpleITK (the reference's arithmetic) is not installable in the build image, so the sitk-facing plumbing of the product
(platipy_amd.image.from_sitk / to_sitk / as_image, and HipDemonsFilter used as the `registration_algorithm` of a
multiscale loop written against the sitk API) would otherwise never execute anywhere.  This module implements just the
calls that plumbing and tests/test_sitk_seam.py's restatement of the reference's loop make, on numpy arrays, with the
voxel-level operations delegated to the CPU oracle (oracle/).  It is NOT a build of the reference, is never on the
product's import path (tests put tests/sitk_double on sys.path explicitly) and says nothing about parity with ITK."""
import numpy as np

from oracle import oracle as _O

__version__ = "test-double"

sitkNearestNeighbor, sitkLinear, sitkBSpline = 1, 2, 3
sitkUInt8, sitkInt16, sitkInt32, sitkInt64, sitkFloat32, sitkFloat64 = 1, 2, 4, 6, 8, 9
sitkVectorFloat32, sitkVectorFloat64 = 21, 22
_SCALAR_DTYPE = {sitkUInt8: np.uint8, sitkInt16: np.int16, sitkInt32: np.int32, sitkInt64: np.int64, sitkFloat32: np.float32,
                 sitkFloat64: np.float64}
_ID_OF = {np.dtype(v): k for k, v in _SCALAR_DTYPE.items()}
_IDENT = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)


class Image:
    """Scalar images hold [Z, Y, X]; vector images hold [Z, Y, X, 3] (the layout GetArrayFromImage returns)."""

    def __init__(self, *args):
        if len(args) == 4 and all(isinstance(a, (int, np.integer)) for a in args):
            w, h, d, pid = args
            if pid in (sitkVectorFloat32, sitkVectorFloat64):
                self._a = np.zeros((d, h, w, 3), np.float64 if pid == sitkVectorFloat64 else np.float32)
            else:
                self._a = np.zeros((d, h, w), _SCALAR_DTYPE[pid])
            self._vec = pid in (sitkVectorFloat32, sitkVectorFloat64)
        elif len(args) == 2:
            self._a, self._vec = args
        else:
            raise TypeError("test double: unsupported Image constructor")
        self._spacing, self._origin, self._direction = (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), _IDENT

    def GetSize(self):
        s = self._a.shape
        return (int(s[2]), int(s[1]), int(s[0]))

    def GetWidth(self):
        return self.GetSize()[0]

    def GetHeight(self):
        return self.GetSize()[1]

    def GetDepth(self):
        return self.GetSize()[2]

    def GetDimension(self):
        return 3

    def GetSpacing(self):
        return self._spacing

    def GetOrigin(self):
        return self._origin

    def GetDirection(self):
        return self._direction

    def SetSpacing(self, s):
        self._spacing = tuple(float(v) for v in s)

    def SetOrigin(self, o):
        self._origin = tuple(float(v) for v in o)

    def SetDirection(self, d):
        self._direction = tuple(float(v) for v in d)

    def CopyInformation(self, other):
        if tuple(other.GetSize()) != self.GetSize():
            raise RuntimeError("CopyInformation: sizes differ")
        self._spacing, self._origin, self._direction = other.GetSpacing(), other.GetOrigin(), other.GetDirection()

    def GetNumberOfComponentsPerPixel(self):
        return 3 if self._vec else 1

    def GetPixelID(self):
        if self._vec:
            return sitkVectorFloat64 if self._a.dtype == np.float64 else sitkVectorFloat32
        return _ID_OF[self._a.dtype]

    GetPixelIDValue = GetPixelID

    # pixel arithmetic: ITK's functors compute in double and store the image's pixel type; a Python scalar keeps the image's type
    def _binary(self, other, fn):
        b = other._a if isinstance(other, Image) else other
        r = fn(self._a.astype(np.float64) if np.issubdtype(self._a.dtype, np.floating) else self._a,
               b.astype(np.float64) if isinstance(b, np.ndarray) and np.issubdtype(b.dtype, np.floating) else b)
        out = Image(np.asarray(r).astype(self._a.dtype), self._vec)
        out.CopyInformation(self)
        return out

    def __add__(self, other):
        return self._binary(other, lambda a, b: a + b)

    __radd__ = __add__

    def __mul__(self, other):
        return self._binary(other, lambda a, b: a * b)

    __rmul__ = __mul__

    def __truediv__(self, other):
        return self._binary(other, lambda a, b: a / b)

    def __pow__(self, other):
        return self._binary(other, lambda a, b: a ** b)

    def __eq__(self, other):      # -> a uint8 image of 0 / 1, as sitk's comparison operators
        b = other._a if isinstance(other, Image) else other
        out = Image((self._a == b).astype(np.uint8), False)
        out.CopyInformation(self)
        return out

    __hash__ = None


def _like(arr, ref, vec):
    out = Image(arr, vec)
    out.SetSpacing(ref.GetSpacing())
    out.SetOrigin(ref.GetOrigin())
    out.SetDirection(ref.GetDirection())
    return out


def GetArrayFromImage(image):
    return image._a.copy()


def GetArrayViewFromImage(image):
    return image._a


def GetImageFromArray(arr, isVector=None):
    a = np.ascontiguousarray(arr)
    vec = bool(isVector) if isVector is not None else False
    if vec and (a.ndim != 4 or a.shape[-1] != 3):
        raise RuntimeError("test double: vector images are [Z, Y, X, 3]")
    return Image(a, vec)


def Cast(image, pixel_id):
    if pixel_id in (sitkVectorFloat32, sitkVectorFloat64):
        if not image._vec:
            raise RuntimeError("Cast: scalar -> vector is not supported")
        return _like(image._a.astype(np.float64 if pixel_id == sitkVectorFloat64 else np.float32), image, True)
    a = image._a
    dt = _SCALAR_DTYPE[pixel_id]
    if np.issubdtype(a.dtype, np.floating) and not np.issubdtype(dt, np.floating):
        a = np.trunc(a)
    return _like(a.astype(dt), image, False)


class Transform:
    def __init__(self, *a):
        pass


class DisplacementFieldTransform(Transform):
    def __init__(self, field_image):
        if not isinstance(field_image, Image) or field_image.GetPixelID() != sitkVectorFloat64:
            raise RuntimeError("DisplacementFieldTransform needs a VectorFloat64 image")   # as SimpleITK insists
        self._field = field_image

    def GetDisplacementField(self):
        return self._field


def _vol(image):
    return _O.Vol(image._a if not image._vec else np.ascontiguousarray(np.moveaxis(image._a, -1, 0)), image.GetSpacing(),
                  image.GetOrigin(), image.GetDirection())


def _ref_vol(size, spacing, origin, direction):
    return _O.Vol(np.zeros((size[2], size[1], size[0]), np.float32), spacing, origin, direction)


def Resample(image, *args):
    """The call shapes the reference uses: (image, reference_image[, transform, interpolator, default]), (image, transform[, interpolator]) and
    (image, size, transform, interpolator, origin, spacing, direction, default_value, pixel_id)."""
    transform, interp, default = None, sitkLinear, 0.0
    if len(args) >= 1 and isinstance(args[0], Image):      # (image, reference_image[, transform[, interpolator[, default]]])
        ref = _ref_vol(args[0].GetSize(), args[0].GetSpacing(), args[0].GetOrigin(), args[0].GetDirection())
        ref_img = args[0]
        if len(args) > 1:
            transform = args[1]
        if len(args) > 2:
            interp = args[2]
        if len(args) > 3:
            default = args[3]
    elif len(args) >= 1 and isinstance(args[0], Transform):
        transform = args[0]
        interp = args[1] if len(args) > 1 else sitkLinear
        ref, ref_img = _ref_vol(image.GetSize(), image.GetSpacing(), image.GetOrigin(), image.GetDirection()), image
    elif len(args) >= 7:
        size, transform, interp, origin, spacing, direction = args[:6]
        default = args[6] if len(args) > 6 else 0.0
        ref = _ref_vol(size, spacing, origin, direction)
        ref_img = Image(np.zeros((size[2], size[1], size[0]), np.float32), False)
        ref_img.SetSpacing(spacing); ref_img.SetOrigin(origin); ref_img.SetDirection(direction)
    else:
        raise TypeError("test double: unsupported Resample call")
    field = None
    if isinstance(transform, DisplacementFieldTransform):
        field = _vol(transform.GetDisplacementField())
    elif transform is not None and type(transform) is not Transform:
        raise TypeError("test double: only identity and displacement-field transforms")
    if image._vec:
        out = _O.resample_vec(_O.Vol(_vol(image).arr.astype(np.float64), image.GetSpacing(), image.GetOrigin(), image.GetDirection()),
                              ref, through=field)
        return _like(np.ascontiguousarray(np.moveaxis(out.arr, 0, -1)).astype(image._a.dtype), ref_img, True)
    out = _O.resample(_vol(image), ref, field_vol=field, interp=_O.INTERP_NEAREST if interp == sitkNearestNeighbor else _O.INTERP_LINEAR,
                      default_value=default)
    return _like(out.arr, ref_img, False)


def DiscreteGaussian(image, variance, maximumKernelWidth=32, maximumError=0.01, useImageSpacing=True):
    out = _O.discrete_gaussian(_O.Vol(image._a.astype(np.float32), image.GetSpacing(), image.GetOrigin(), image.GetDirection()), variance,
                               maximumKernelWidth, maximumError, useImageSpacing)
    return _like(out.arr, image, False)


def SmoothingRecursiveGaussian(image, sigma):
    sig = [float(v) for v in np.broadcast_to(np.asarray(sigma, dtype=np.float64), (3,))]
    if image._vec:
        out = _O.recursive_gaussian_vec(_O.Vol(_vol(image).arr.astype(np.float64), image.GetSpacing(), image.GetOrigin(), image.GetDirection()), sig)
        return _like(np.ascontiguousarray(np.moveaxis(out.arr, 0, -1)), image, True)
    return _like(_O.recursive_gaussian(_vol(image), sig).arr, image, False)


def Version_VersionString():
    return __version__


class Version:
    VersionString = staticmethod(Version_VersionString)


class FastSymmetricForcesDemonsRegistrationFilter:
    """The filter protocol of deformable.py:244-257,149 on the oracle's restatement (SimpleITK 2.3.1 defaults)."""

    def __init__(self):
        self._f = _O.DemonsFilter()

    def SetNumberOfIterations(self, n):
        self._f.SetNumberOfIterations(n)

    def SetStandardDeviations(self, s):
        self._f.SetStandardDeviations(s)

    def GetStandardDeviations(self):
        return self._f.GetStandardDeviations()

    def SetSmoothUpdateField(self, b):
        self._f.SetSmoothUpdateField(b)

    def SetSmoothDisplacementField(self, b):
        self._f.SetSmoothDisplacementField(b)

    def SetMaximumRMSError(self, v):
        self._f.SetMaximumRMSError(v)

    def SetNumberOfThreads(self, n):
        pass

    def GetElapsedIterations(self):
        return self._f.GetElapsedIterations()

    def GetMetric(self):
        return self._f.GetMetric()

    def GetRMSChange(self):
        return self._f.GetRMSChange()

    def Execute(self, fixed, moving):
        out = self._f.Execute(_vol(fixed), _vol(moving))
        return _like(np.ascontiguousarray(np.moveaxis(out.arr, 0, -1)).astype(np.float64), fixed, True)


def SignedMaurerDistanceMap(image, insideIsPositive=False, squaredDistance=True, useImageSpacing=False):
    if squaredDistance or not useImageSpacing:
        raise TypeError("test double: only squaredDistance=False, useImageSpacing=True (the reference's call, projection.py:80-82)")
    return _like(_O.maurer_distance_map(_vol(image), signed=True, inside_positive=bool(insideIsPositive)).arr.astype(np.float32), image, False)


def LabelContour(image, fullyConnected=False, backgroundValue=0):
    return _like(_O.label_contour(_vol(image)).arr.astype(image._a.dtype), image, False)


# ---- round 4: what tools/sitk_vectors.py's remaining stages call (fusion.py:148-190, 263-328; registration/utils.py:216-267,
# 328-329; registration/linear.py:133-153).  Backed by the oracle / numpy like everything here: plumbing, not ITK.
sitkBall = 1


def SquaredDifference(a, b):
    d = a._a.astype(np.float64) - b._a.astype(np.float64)
    return _like((d * d).astype(a._a.dtype), a, False)


def Pow(image, exponent):
    return _like((image._a.astype(np.float64) ** float(exponent)).astype(image._a.dtype), image, False)


def BoxMean(image, radius):
    from scipy.ndimage import uniform_filter

    r = (radius,) * 3 if isinstance(radius, int) else tuple(radius)
    out = uniform_filter(image._a.astype(np.float64), size=[2 * b + 1 for b in r[::-1]], mode="nearest")
    return _like(out.astype(image._a.dtype), image, False)


def Mask(image, mask, outsideValue=0, maskingValue=0):
    return _like(np.where(mask._a == maskingValue, np.asarray(outsideValue, dtype=image._a.dtype), image._a), image, False)


def RescaleIntensity(image, outputMinimum=0, outputMaximum=255):
    return _like(_O.rescale_intensity(image._a, float(outputMinimum), float(outputMaximum)).astype(image._a.dtype), image, False)


def Threshold(image, lower=0.0, upper=1.0, outsideValue=0.0):
    a = image._a
    return _like(np.where((a < np.asarray(lower, a.dtype)) | (a > np.asarray(upper, a.dtype)), np.asarray(outsideValue, a.dtype), a), image, False)


def BinaryThreshold(image, lowerThreshold=0.0, upperThreshold=255.0, insideValue=1, outsideValue=0):
    a = image._a.astype(np.float64)
    return _like(np.where((a >= lowerThreshold) & (a <= upperThreshold), insideValue, outsideValue).astype(np.uint8), image, False)


def BinaryFillhole(image, fullyConnected=False, foregroundValue=1):
    from scipy import ndimage

    return _like(ndimage.binary_fill_holes(image._a == foregroundValue).astype(image._a.dtype), image, False)


def ConnectedComponent(image, fullyConnected=False):
    from scipy import ndimage

    lab, _ = ndimage.label(image._a != 0)          # face connectivity, components numbered in raster order of their first voxel
    return _like(lab.astype(np.uint32), image, False)


def BinaryDilate(image, radius, kernel=sitkBall, *a, **k):
    return _like(_O.binary_dilate_ball(_vol(image), tuple(radius)).arr.astype(image._a.dtype), image, False)


def BinaryMorphologicalClosing(image, radius, kernel=sitkBall, *a, **k):
    return _like(_O.binary_closing_ball(_vol(image), tuple(radius)).arr.astype(image._a.dtype), image, False)


class AffineTransform(Transform):
    def __init__(self, dim=3):
        self._A, self._t, self._c = np.eye(3), np.zeros(3), np.zeros(3)

    def SetCenter(self, c):
        self._c = np.asarray(c, dtype=np.float64)

    def SetMatrix(self, m):
        self._A = np.asarray(m, dtype=np.float64).reshape(3, 3)

    def SetTranslation(self, t):
        self._t = np.asarray(t, dtype=np.float64)


def _not_here(what):
    raise NotImplementedError(f"test double: {what} needs the real SimpleITK")


def CenteredTransformInitializer(*a, **k):
    _not_here("CenteredTransformInitializer")


class ImageRegistrationMethod:
    """MetricEvaluate for mean squares over every voxel of the fixed grid at an AffineTransform (registration/linear.py:133-153's
    metric at sampling 1.0); the optimiser (Execute) is the real library's business."""
    NONE, REGULAR, RANDOM = 0, 1, 2

    def __init__(self):
        self._tfm, self._fixed_mask = None, None

    def SetMetricAsMeanSquares(self):
        pass

    def SetMetricFixedMask(self, mask):
        self._fixed_mask = mask

    def SetMetricSamplingStrategy(self, s):
        if s != self.NONE:
            _not_here("a sampled metric")

    def SetInterpolator(self, i):
        if i != sitkLinear:
            _not_here("a non-linear interpolator in the metric")

    def SetInitialTransform(self, t, inPlace=True):
        self._tfm = t

    def MetricEvaluate(self, fixed, moving):
        from oracle import linear_oracle

        T = self._tfm
        if not isinstance(T, AffineTransform):
            _not_here("MetricEvaluate on this transform")
        sf, sm = np.array(fixed.GetSpacing()), np.array(moving.GetSpacing())
        of, om = np.array(fixed.GetOrigin()), np.array(moving.GetOrigin())
        Am = (T._A * sf[None, :]) / sm[:, None]
        bm = (T._A @ (of - T._c) + T._c + T._t - om) / sm
        r = linear_oracle.meansq_affine(fixed._a, moving._a, np.eye(3), np.zeros(3), Am, bm, fixed.GetSize(), 1,
                                        fixed_mask=None if self._fixed_mask is None else self._fixed_mask._a)
        return float(r[0] / r[1])

    def Execute(self, *a, **k):
        _not_here("ImageRegistrationMethod.Execute")
