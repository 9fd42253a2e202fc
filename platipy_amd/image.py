"""A volume in HBM plus its physical geometry -- the stand-in for sitk.Image on this path.

`tensor` is [Z, Y, X] for scalar images and [3, Z, Y, X] (planar) for displacement fields, x
fastest, exactly the memory order of sitk.GetArrayFromImage for scalars.  Accessor names follow
SimpleITK so code written against the reference reads the same.
"""
import numpy as np
import torch

from . import _lib

_IDENTITY = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)


def _default_device():
    from . import runtime

    return runtime.default_device()


class Image:
    def __init__(self, tensor, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=_IDENTITY, is_vector=None):
        if isinstance(tensor, np.ndarray):
            tensor = torch.from_numpy(np.ascontiguousarray(tensor)).to(_default_device())
        if is_vector is None:
            is_vector = tensor.dim() == 4
        if is_vector and (tensor.dim() != 4 or tensor.shape[0] != 3):
            raise ValueError("vector images are planar [3, Z, Y, X]")
        if not is_vector and tensor.dim() != 3:
            raise ValueError("scalar images are [Z, Y, X]")
        self.tensor = tensor.contiguous()
        self.is_vector = bool(is_vector)
        self.spacing = tuple(float(s) for s in spacing)
        self.origin = tuple(float(o) for o in origin)
        self.direction = tuple(float(d) for d in direction)
        if len(self.spacing) != 3 or len(self.origin) != 3 or len(self.direction) != 9:
            raise ValueError("3-D geometry expected")

    # -- SimpleITK-style accessors --------------------------------------------------
    def GetSize(self):
        s = self.tensor.shape[-3:]
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
        return self.spacing

    def GetOrigin(self):
        return self.origin

    def GetDirection(self):
        return self.direction

    def GetNumberOfComponentsPerPixel(self):
        return 3 if self.is_vector else 1

    def GetPixelID(self):
        return self.tensor.dtype

    GetPixelIDValue = GetPixelID

    def SetSpacing(self, s):
        self.spacing = tuple(float(v) for v in s)

    def SetOrigin(self, o):
        self.origin = tuple(float(v) for v in o)

    def SetDirection(self, d):
        self.direction = tuple(float(v) for v in d)

    def CopyInformation(self, other):
        if tuple(other.GetSize()) != tuple(self.GetSize()):
            raise ValueError("CopyInformation: sizes differ")
        self.spacing, self.origin, self.direction = other.spacing, other.origin, other.direction

    # -- helpers --------------------------------------------------------------------
    @property
    def device(self):
        return self.tensor.device

    @property
    def shape(self):
        return tuple(self.tensor.shape[-3:])

    def like(self, tensor, is_vector=None):
        return Image(tensor, self.spacing, self.origin, self.direction, is_vector)

    def same_grid(self, other):
        return (self.GetSize() == other.GetSize() and self.spacing == other.spacing and self.origin == other.origin
                and self.direction == other.direction)

    def geom(self):
        return _lib.make_geom(self.GetSize(), self.spacing, self.origin, self.direction)

    def astype(self, dtype):
        return self.like(cast_tensor(self.tensor, dtype), self.is_vector)

    def numpy(self):
        return self.tensor.detach().cpu().numpy()

    def as_interleaved(self):
        """[Z, Y, X, 3] float64 array, the layout sitk.GetArrayFromImage gives for a VectorFloat64 field."""
        if not self.is_vector:
            raise ValueError("not a vector image")
        return self.tensor.permute(1, 2, 3, 0).contiguous().double().cpu().numpy()

    def __repr__(self):
        kind = "vector" if self.is_vector else "scalar"
        return f"Image({kind} {self.GetSize()} {self.tensor.dtype} spacing={self.spacing} origin={self.origin})"


def cast_tensor(t, dtype):
    """sitk.Cast semantics: float -> integer truncates toward zero (C++ static_cast)."""
    if t.dtype == dtype:
        return t
    if t.dtype.is_floating_point and not dtype.is_floating_point:
        return torch.trunc(t).to(dtype)
    return t.to(dtype)


def image_from_array(arr, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=_IDENTITY, device=None, is_vector=None):
    """sitk.GetImageFromArray + geometry.  `arr` is [Z, Y, X] (or planar [3, Z, Y, X] with is_vector)."""
    a = np.ascontiguousarray(arr)
    if a.dtype == np.float64 and not is_vector:
        pass  # kept as float64 until an operation casts it, like sitkFloat64
    t = torch.from_numpy(a).to(device or _default_device())
    return Image(t, spacing, origin, direction, is_vector if is_vector is not None else False)


def array_from_image(image):
    return image.numpy()


def from_sitk(image, device=None):
    """Convert a SimpleITK image (only if SimpleITK is installed where this runs)."""
    import SimpleITK as sitk

    arr = sitk.GetArrayFromImage(image)
    vec = image.GetNumberOfComponentsPerPixel() == 3
    if vec:
        arr = np.ascontiguousarray(np.moveaxis(arr, -1, 0)).astype(np.float32)
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(device or _default_device())
    return Image(t, image.GetSpacing(), image.GetOrigin(), image.GetDirection(), vec)


def to_sitk(image):
    import SimpleITK as sitk

    if image.is_vector:
        out = sitk.GetImageFromArray(image.as_interleaved(), isVector=True)
    else:
        out = sitk.GetImageFromArray(image.numpy())
    out.SetSpacing(image.spacing)
    out.SetOrigin(image.origin)
    out.SetDirection(image.direction)
    return out


def as_image(x, device=None):
    """Accept platipy_amd.Image or (when SimpleITK exists) sitk.Image."""
    if isinstance(x, Image):
        return x
    if type(x).__module__.startswith("SimpleITK"):
        return from_sitk(x, device)
    raise TypeError(f"expected platipy_amd.Image (or sitk.Image), got {type(x)!r}")
