"""NIfTI-1 reader/writer: round trips and a hand-built header (geometry in ITK's LPS convention)."""
import gzip
import struct

import numpy as np
import pytest
import torch

from platipy_amd import io as pio
from platipy_amd.image import Image


@pytest.mark.parametrize("dtype", [np.float32, np.int16, np.uint8])
@pytest.mark.parametrize("ext", [".nii", ".nii.gz"])
def test_round_trip(tmp_path, dtype, ext):
    rng = np.random.default_rng(0)
    arr = (rng.normal(size=(5, 7, 9)) * 100).astype(dtype)
    ang = 0.3
    d = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
    img = Image(torch.from_numpy(arr), (0.9, 1.1, 2.5), (320.0, -52.0, 60.0), tuple(d.ravel()))
    p = tmp_path / f"v{ext}"
    pio.write_image(img, p)
    back = pio.read_image(p, device="cpu")
    assert back.tensor.dtype == img.tensor.dtype and back.GetSize() == (9, 7, 5)
    np.testing.assert_array_equal(back.numpy(), arr)
    np.testing.assert_allclose(back.spacing, img.spacing, rtol=1e-6)
    np.testing.assert_allclose(back.origin, img.origin, rtol=1e-6)
    np.testing.assert_allclose(back.direction, img.direction, atol=1e-6)


def test_reads_plain_ras_header(tmp_path):
    """An identity-RAS file (what most tools write) is an LPS image with direction diag(-1, -1, 1)."""
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, 3, 4, 3, 2, 1, 1, 1, 1)
    struct.pack_into("<h", hdr, 70, 16)
    struct.pack_into("<h", hdr, 72, 32)
    struct.pack_into("<8f", hdr, 76, 1, 2.0, 3.0, 4.0, 0, 0, 0, 0)
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<2h", hdr, 252, 0, 2)
    for r, row in enumerate([(2.0, 0, 0, 10.0), (0, 3.0, 0, 20.0), (0, 0, 4.0, 30.0)]):
        struct.pack_into("<4f", hdr, 280 + 16 * r, *row)
    hdr[344:348] = b"n+1\0"
    data = np.arange(24, dtype="<f4")
    p = tmp_path / "ras.nii.gz"
    with gzip.open(p, "wb") as f:
        f.write(bytes(hdr) + b"\0\0\0\0" + data.tobytes())
    img = pio.read_image(p, device="cpu")
    assert img.GetSize() == (4, 3, 2)
    np.testing.assert_allclose(img.spacing, (2.0, 3.0, 4.0))
    np.testing.assert_allclose(img.origin, (-10.0, -20.0, 30.0))
    np.testing.assert_allclose(np.array(img.direction).reshape(3, 3), np.diag([-1.0, -1.0, 1.0]))
    assert img.numpy()[1, 2, 3] == 23.0
    with pytest.raises(ValueError):
        (tmp_path / "bad.nii").write_bytes(b"\0" * 400)
        pio.read_image(tmp_path / "bad.nii", device="cpu")
