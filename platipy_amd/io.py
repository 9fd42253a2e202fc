"""Minimal NIfTI-1 (.nii / .nii.gz) reader and writer, so the multi-atlas pipeline runs file-to-file without
SimpleITK (reference: sitk.ReadImage at projects/multiatlas/run.py:160-165, sitk.WriteImage at
cli/segmentation.py:127-128).  Scalar 3-D volumes only; geometry follows ITK's NIfTI convention: the file's
affine is RAS, images are LPS, so the first two rows change sign; spacing = column norms, direction =
normalised columns, origin = translation."""
import gzip
import struct

import numpy as np
import torch

from .image import Image

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32}
_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def _quat_to_mat(b, c, d, qfac, pixdim):
    a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    return R * np.array([pixdim[0], pixdim[1], pixdim[2] * qfac])[None, :]


def read_image(path, device=None):
    """sitk.ReadImage for a 3-D scalar NIfTI-1 file -> Image resident on `device` (default: current GPU)."""
    with _open(path, "rb") as f:
        raw = f.read()
    hdr = raw[:348]
    for end in ("<", ">"):
        if struct.unpack(end + "i", hdr[0:4])[0] == 348:
            break
    else:
        raise ValueError(f"{path}: not a NIfTI-1 file")
    dim = struct.unpack(end + "8h", hdr[40:56])
    if dim[0] < 3 or any(d > 1 for d in dim[4:dim[0] + 1]):
        raise ValueError(f"{path}: only 3-D scalar volumes are supported (dim = {dim})")
    nx, ny, nz = dim[1:4]
    datatype = struct.unpack(end + "h", hdr[70:72])[0]
    if datatype not in _DTYPES:
        raise ValueError(f"{path}: unsupported datatype code {datatype}")
    pixdim = struct.unpack(end + "8f", hdr[76:108])
    vox_offset = int(struct.unpack(end + "f", hdr[108:112])[0])
    slope, inter = struct.unpack(end + "2f", hdr[112:120])
    qform_code, sform_code = struct.unpack(end + "2h", hdr[252:256])
    if sform_code > 0:
        A = np.array([struct.unpack(end + "4f", hdr[280 + 16 * r:296 + 16 * r]) for r in range(3)], dtype=np.float64)
        M, t = A[:, :3], A[:, 3]
    elif qform_code > 0:
        b, c, d, qx, qy, qz = struct.unpack(end + "6f", hdr[256:280])
        qfac = -1.0 if pixdim[0] < 0 else 1.0
        M, t = _quat_to_mat(b, c, d, qfac, pixdim[1:4]), np.array([qx, qy, qz], dtype=np.float64)
    else:
        M, t = np.diag(np.abs(pixdim[1:4])).astype(np.float64), np.zeros(3)
    lps = np.diag([-1.0, -1.0, 1.0])
    M, t = lps @ M, lps @ t
    spacing = np.sqrt((M ** 2).sum(0))
    direction = M / spacing[None, :]
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    arr = np.frombuffer(raw, dtype=dt, count=nx * ny * nz, offset=max(vox_offset, 352)).reshape(nz, ny, nx)
    arr = arr.astype(dt.newbyteorder("="))
    if slope not in (0.0, 1.0) or (slope != 0.0 and inter != 0.0):
        arr = (arr.astype(np.float64) * slope + inter).astype(np.float32)
    from . import runtime

    ten = torch.from_numpy(np.ascontiguousarray(arr)).to(device or runtime.default_device())
    return Image(ten, tuple(spacing), tuple(t), tuple(direction.ravel()))


def write_image(image, path):
    """sitk.WriteImage for a 3-D scalar Image -> NIfTI-1 (sform and qform both set)."""
    if image.is_vector:
        raise ValueError("vector images are not supported")
    arr = np.ascontiguousarray(image.numpy())
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    if arr.dtype not in _CODES:
        arr = arr.astype(np.float32)
    nz, ny, nx = arr.shape
    d = np.asarray(image.direction, dtype=np.float64).reshape(3, 3)
    sp = np.asarray(image.spacing, dtype=np.float64)
    lps = np.diag([-1.0, -1.0, 1.0])
    M = lps @ (d * sp[None, :])
    t = lps @ np.asarray(image.origin, dtype=np.float64)
    R = M / sp[None, :]
    qfac = 1.0
    if np.linalg.det(R) < 0:
        R = R.copy()
        R[:, 2] *= -1
        qfac = -1.0
    a = 0.5 * np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2]))
    if a > 1e-6:
        b, c, dd = (R[2, 1] - R[1, 2]) / (4 * a), (R[0, 2] - R[2, 0]) / (4 * a), (R[1, 0] - R[0, 1]) / (4 * a)
    else:   # 180-degree rotations
        b = np.sqrt(max(0.0, (1 + R[0, 0]) / 2))
        c = np.sqrt(max(0.0, (1 + R[1, 1]) / 2)) * (1 if R[0, 1] >= 0 else -1)
        dd = np.sqrt(max(0.0, (1 + R[2, 2]) / 2)) * (1 if R[0, 2] >= 0 else -1)
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into("<h", hdr, 70, _CODES[arr.dtype])
    struct.pack_into("<h", hdr, 72, arr.dtype.itemsize * 8)
    struct.pack_into("<8f", hdr, 76, qfac, sp[0], sp[1], sp[2], 0, 0, 0, 0)
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<2f", hdr, 112, 1.0, 0.0)
    hdr[123] = 2  # xyzt_units: mm
    struct.pack_into("<2h", hdr, 252, 1, 1)
    struct.pack_into("<6f", hdr, 256, b, c, dd, t[0], t[1], t[2])
    for r in range(3):
        struct.pack_into("<4f", hdr, 280 + 16 * r, M[r, 0], M[r, 1], M[r, 2], t[r])
    hdr[344:348] = b"n+1\0"
    with _open(path, "wb") as f:
        f.write(bytes(hdr) + b"\0\0\0\0" + arr.astype(arr.dtype.newbyteorder("<")).tobytes())
