"""Transforms understood by apply_transform -- the subset of sitk.Transform the path uses."""
import numpy as np

# interpolator constants with SimpleITK's values
sitkNearestNeighbor = 1
sitkLinear = 2
sitkBSpline = 3


class Transform:
    """Identity (sitk.Transform())."""

    def matrix_offset(self):
        return np.eye(3), np.zeros(3)

    def is_linear(self):
        return True


class AffineTransform(Transform):
    """q = A (p - c) + c + t, itk::MatrixOffsetTransformBase (rigid / similarity / affine all reduce to it)."""

    def __init__(self, matrix=None, translation=(0.0, 0.0, 0.0), center=(0.0, 0.0, 0.0)):
        self.matrix = np.eye(3) if matrix is None else np.asarray(matrix, dtype=np.float64).reshape(3, 3)
        self.translation = np.asarray(translation, dtype=np.float64).reshape(3)
        self.center = np.asarray(center, dtype=np.float64).reshape(3)

    def matrix_offset(self):
        off = self.translation + self.center - self.matrix @ self.center
        return self.matrix.copy(), off

    def GetMatrix(self):
        return tuple(self.matrix.ravel())

    def GetTranslation(self):
        return tuple(self.translation)

    def GetCenter(self):
        return tuple(self.center)

    def __repr__(self):
        return f"AffineTransform(matrix={self.matrix.tolist()}, translation={self.translation.tolist()}, center={self.center.tolist()})"


class DisplacementFieldTransform(Transform):
    """q = p + D(p) with D a planar vector Image (sitk.DisplacementFieldTransform)."""

    def __init__(self, field_image):
        if not getattr(field_image, "is_vector", False):
            raise ValueError("DisplacementFieldTransform needs a vector Image")
        self.field = field_image

    def GetDisplacementField(self):
        return self.field

    def is_linear(self):
        return False


class CompositeTransform(Transform):
    """sitk.CompositeTransform([t0, t1, ...]): T(p) = t0(t1(...(p))) -- the LAST transform is applied first."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def AddTransform(self, t):
        self.transforms.append(t)

    def is_linear(self):
        return all(t.is_linear() for t in self.transforms)

    def matrix_offset(self):
        A, off = np.eye(3), np.zeros(3)
        for t in reversed(self.transforms):  # apply last first
            a, o = t.matrix_offset()
            A, off = a @ A, a @ off + o
        return A, off

    def flatten(self):
        out = []
        for t in self.transforms:
            out.extend(t.flatten() if isinstance(t, CompositeTransform) else [t])
        return out


# --------------------------------------------------------------------------------------
# parametrised linear transforms (the models linear_registration optimises, reference
# registration/linear.py:166-181).  All are MatrixOffsetTransformBase: q = A (p - c) + c + t.


def _versor_update(p, update):
    """itk::VersorRigid3DTransform::UpdateTransformParameters (inherited by Similarity3D, ScaleVersor3D and ScaleSkewVersor3D):
    the first three entries of `update` are an AXIS-ANGLE rotation -- axis update[:3], angle |update[:3]| radians -- composed
    onto the current rotation on the right (new = current * gradientRotation, Hamilton product); every other parameter is
    added.  A zero update leaves the versor as it is."""
    p = np.asarray(p, dtype=np.float64)
    update = np.asarray(update, dtype=np.float64)
    out = p + update
    x, y, z = p[0], p[1], p[2]
    w = np.sqrt(max(0.0, 1.0 - (x * x + y * y + z * z)))
    norm = float(np.sqrt(update[0] * update[0] + update[1] * update[1] + update[2] * update[2]))
    if norm > 0.0:
        f = np.sin(norm / 2.0) / norm
        gx, gy, gz, gw = update[0] * f, update[1] * f, update[2] * f, np.cos(norm / 2.0)
    else:
        gx, gy, gz, gw = 0.0, 0.0, 0.0, 1.0
    nx = w * gx - z * gy + y * gz + x * gw
    ny = z * gx + w * gy - x * gz + y * gw
    nz = -y * gx + x * gy + w * gz + z * gw
    nw = -x * gx - y * gy - z * gz + w * gw
    if nw < 0.0:      # (itk::Versor keeps w >= 0 implicitly: only the right part is stored and w = +sqrt(1 - |v|^2))
        nx, ny, nz = -nx, -ny, -nz
    out[0], out[1], out[2] = nx, ny, nz
    return out


def _versor_matrix(v):
    x, y, z = (float(a) for a in v)
    n2 = x * x + y * y + z * z
    if n2 > 1.0:  # (rounding only: composed versors have unit norm)
        s = 1.0 / np.sqrt(n2)
        x, y, z = x * s, y * s, z * s
        n2 = 1.0
    w = np.sqrt(max(0.0, 1.0 - n2))
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


class _Parametrised(AffineTransform):
    n_params = 0

    def __init__(self, center=(0.0, 0.0, 0.0)):
        super().__init__(np.eye(3), (0.0, 0.0, 0.0), center)
        self.SetParameters(self.identity_parameters())

    def identity_parameters(self):
        raise NotImplementedError

    def decode(self, params):
        """-> (A, t) for a parameter vector."""
        raise NotImplementedError

    def GetParameters(self):
        return tuple(self._params)

    def SetParameters(self, params):
        self._params = np.asarray(params, dtype=np.float64).copy()
        self.matrix, self.translation = self.decode(self._params)

    def GetNumberOfParameters(self):
        return self.n_params

    def update(self, params, update):
        """itk::Transform::UpdateTransformParameters(update, factor = 1) on a parameter vector: params + update, except
        where a transform class overrides it (the versor family composes its rotation, _versor_update)."""
        return np.asarray(params, dtype=np.float64) + np.asarray(update, dtype=np.float64)

    def SetCenter(self, c):
        self.center = np.asarray(c, dtype=np.float64).reshape(3)


class TranslationTransform(_Parametrised):
    n_params = 3

    def identity_parameters(self):
        return np.zeros(3)

    def decode(self, p):
        return np.eye(3), np.asarray(p[:3], dtype=np.float64).copy()


class VersorRigid3DTransform(_Parametrised):
    """parameters: versor (x, y, z), translation (3)"""
    n_params = 6

    def identity_parameters(self):
        return np.zeros(6)

    def decode(self, p):
        return _versor_matrix(p[:3]), np.asarray(p[3:6], dtype=np.float64).copy()

    update = staticmethod(_versor_update)


class Similarity3DTransform(_Parametrised):
    """parameters: versor (x, y, z), translation (3), isotropic scale"""
    n_params = 7

    def identity_parameters(self):
        return np.array([0, 0, 0, 0, 0, 0, 1.0])

    def decode(self, p):
        return float(p[6]) * _versor_matrix(p[:3]), np.asarray(p[3:6], dtype=np.float64).copy()

    update = staticmethod(_versor_update)


class ScaleVersor3DTransform(_Parametrised):
    """itk::ScaleVersor3DTransform -- parameters: versor (x, y, z), translation (3), scale (3).  ITK does not compose
    rotation and scaling here: the matrix is the rotation matrix with (scale_i - 1) ADDED to its diagonal
    (ScaleVersor3DTransform::ComputeMatrix)."""
    n_params = 9

    def identity_parameters(self):
        return np.array([0, 0, 0, 0, 0, 0, 1.0, 1.0, 1.0])

    def decode(self, p):
        A = _versor_matrix(p[:3])
        A[0, 0] += float(p[6]) - 1.0
        A[1, 1] += float(p[7]) - 1.0
        A[2, 2] += float(p[8]) - 1.0
        return A, np.asarray(p[3:6], dtype=np.float64).copy()

    update = staticmethod(_versor_update)


class ScaleSkewVersor3DTransform(_Parametrised):
    """itk::ScaleSkewVersor3DTransform -- parameters: versor (3), translation (3), scale (3), skew (6).  As in
    ScaleVersor3D the scale and the six skew terms are ADDED to the rotation matrix: diagonal += scale_i - 1, off-diagonal
    entries (0,1) (0,2) (1,0) (1,2) (2,0) (2,1) += skew_0..5 (ScaleSkewVersor3DTransform::ComputeMatrix, additive form).

    Which ITK this reproduces: the 15-parameter class that sitk.ScaleSkewVersor3DTransform wraps in SimpleITK 2.x / ITK 5.x.
    ITK 5 added a SEPARATE class, ComposeScaleSkewVersor3DTransform (12 parameters: three skews, matrix = rotation * scale
    * upper-triangular skew), for the composed form and left this one additive for backward compatibility; the reference's
    reg_method="ScaleSkewVersor" (linear.py:179-180) constructs the 15-parameter class.  No ITK source is available here:
    this is a recollection (DESIGN.md section 3 lists it with the others) and tools/compare_with_sitk.py is where it gets
    checked the day a SimpleITK box exists.  pp_linear.hip's PP_MODEL_SCALE_SKEW_VERSOR decodes identically."""
    n_params = 15

    def identity_parameters(self):
        return np.array([0, 0, 0, 0, 0, 0, 1.0, 1.0, 1.0, 0, 0, 0, 0, 0, 0])

    def decode(self, p):
        A = _versor_matrix(p[:3])
        A[0, 0] += float(p[6]) - 1.0
        A[1, 1] += float(p[7]) - 1.0
        A[2, 2] += float(p[8]) - 1.0
        A[0, 1] += float(p[9]);  A[0, 2] += float(p[10])
        A[1, 0] += float(p[11]); A[1, 2] += float(p[12])
        A[2, 0] += float(p[13]); A[2, 1] += float(p[14])
        return A, np.asarray(p[3:6], dtype=np.float64).copy()

    update = staticmethod(_versor_update)


class ScaleTransform(_Parametrised):
    n_params = 3

    def identity_parameters(self):
        return np.ones(3)

    def decode(self, p):
        return np.diag(np.asarray(p[:3], dtype=np.float64)), np.zeros(3)


class FullAffineTransform(_Parametrised):
    """sitk.AffineTransform(3): parameters = matrix (row-major, 9), translation (3)"""
    n_params = 12

    def identity_parameters(self):
        return np.concatenate([np.eye(3).ravel(), np.zeros(3)])

    def decode(self, p):
        return np.asarray(p[:9], dtype=np.float64).reshape(3, 3).copy(), np.asarray(p[9:12], dtype=np.float64).copy()


class Euler3DTransform(_Parametrised):
    """parameters: angles (x, y, z; ZXY order as ITK's default), translation (3)"""
    n_params = 6

    def identity_parameters(self):
        return np.zeros(6)

    def decode(self, p):
        ax, ay, az = (float(a) for a in p[:3])
        cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
        Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        return Rz @ Rx @ Ry, np.asarray(p[3:6], dtype=np.float64).copy()
