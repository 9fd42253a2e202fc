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
