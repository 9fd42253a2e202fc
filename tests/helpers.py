"""Shared test plumbing: two ways to reach the kernels behind the C ABI, and synthetic volumes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from platipy_amd import _lib  # noqa: E402


class EmuBackend:
    """TEST ONLY: the kernel sources compiled for the CPU (tests/emu); buffers are numpy arrays."""

    name = "emu"

    def __init__(self):
        from tests.emu.build_emu import build

        self.lib = _lib.load(build())
        self.ctx = _lib.Context(0, 0, lib=self.lib)

    def dev(self, a):
        return np.ascontiguousarray(a).copy()

    def host(self, h):
        return np.asarray(h)

    def empty(self, shape, dtype=np.float32):
        return np.zeros(shape, dtype=dtype)


class GpuBackend:
    """The real libplatipy_hip.so on cuda:0; buffers are torch tensors."""

    name = "gpu"

    def __init__(self):
        import torch

        assert torch.cuda.is_available(), "GPU tests need a GPU"
        self.torch = torch
        self.lib = _lib.dll()
        self.ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream, lib=self.lib)

    def dev(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def host(self, h):
        self.ctx.sync()
        return h.cpu().numpy()

    def empty(self, shape, dtype=np.float32):
        tdt = {np.float32: self.torch.float32, np.uint8: self.torch.uint8}[dtype]
        return self.torch.zeros(tuple(shape), dtype=tdt, device="cuda")


# --------------------------------------------------------------------------------------
# synthetic data (SURVEY 8d recipe, scaled down)


def smooth_noise(shape, seed, cells=6):
    """Smooth random scalar field: coarse N(0,1) noise, trilinearly up-sampled."""
    from scipy.ndimage import zoom

    rng = np.random.default_rng(seed)
    coarse = rng.normal(size=(cells, cells, cells))
    z = [s / cells for s in shape]
    out = zoom(coarse, z, order=1, mode="nearest", grid_mode=False)
    out = out[: shape[0], : shape[1], : shape[2]]
    pad = [(0, shape[i] - out.shape[i]) for i in range(3)]
    return np.pad(out, pad, mode="edge")


def phantom(shape, seed=1234, n_blobs=8, noise=5.0):
    """CT-like volume: -1000 background, 0 HU ellipsoid body, ellipsoidal organs, blur, noise."""
    from scipy.ndimage import gaussian_filter

    rng = np.random.default_rng(seed)
    nz, ny, nx = shape
    zz, yy, xx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    vol = np.full(shape, -1000.0)
    body = ((xx - nx / 2) / (0.42 * nx)) ** 2 + ((yy - ny / 2) / (0.40 * ny)) ** 2 + ((zz - nz / 2) / (0.46 * nz)) ** 2 < 1
    vol[body] = 0.0
    for _ in range(n_blobs):
        c = [rng.uniform(0.25, 0.75) * s for s in (nx, ny, nz)]
        r = [rng.uniform(0.06, 0.2) * s for s in (nx, ny, nz)]
        val = rng.uniform(-200, 400)
        m = ((xx - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((zz - c[2]) / r[2]) ** 2 < 1
        vol[m & body] = val
    vol = gaussian_filter(vol, 1.5)
    if noise:
        vol = vol + np.random.default_rng(seed + 1).normal(0, noise, size=shape)
    return vol.astype(np.float32)


def random_dvf(shape, spacing, seed, max_mm=4.0, cells=5):
    f = np.stack([smooth_noise(shape, seed + c, cells) for c in range(3)])
    f *= max_mm / np.sqrt((f ** 2).sum(0)).max()
    return f.astype(np.float32)


def dice(a, b):
    a = a > 0
    b = b > 0
    return 2.0 * (a & b).sum() / max(1, a.sum() + b.sum())


def install_emu_runtime(setattr_fn=None):
    """TEST PLUMBING: point platipy_amd's context lookup at the CPU-emulated kernels (tensors stay on the host).
    linear_registration evaluates the metric thousands of times; emulating each 256-thread reduction launch
    thread-by-thread is too slow for the CPU suite, so here (only) the metric evaluation is the oracle's vectorised
    restatement -- the kernel itself is compared with it in tests/test_linear.py::test_meansq_kernel_*.
    `setattr_fn(obj, name, value)` defaults to plain setattr (used by spawned worker processes)."""
    import torch

    from oracle import linear_oracle
    from platipy_amd import runtime

    be = EmuBackend()
    sa = setattr_fn or setattr
    sa(runtime, "context", lambda device=None: be.ctx)
    sa(runtime, "default_device", lambda: torch.device("cpu"))

    def jit():      # the jitter array the host code set on the context (linear_registration(itk_sampling=True)), or None
        t = getattr(be.ctx, "_sample_jitter", None)
        return None if t is None else (t.numpy() if hasattr(t, "numpy") else np.asarray(t))

    def mgrad():    # ... and the filtered gradient image, or None
        t = getattr(be.ctx, "_moving_gradient", None)
        return None if t is None else (t.numpy() if hasattr(t, "numpy") else np.asarray(t))

    def fake_meansq(fixed, fsize, moving, msize, Af, bf, Am, bm, vsize, stride, fixed_mask=None, moving_mask=None):
        tn = lambda t: None if t is None else t.numpy()  # noqa: E731
        return list(linear_oracle.meansq_affine(fixed.numpy(), moving.numpy(), Af, bf, Am, bm, vsize, stride, tn(fixed_mask),
                                                tn(moving_mask), jitter=jit(), moving_gradient=mgrad()))

    def fake_corr(fixed, fsize, moving, msize, Af, bf, Am, bm, vsize, stride, fixed_mask=None, moving_mask=None):
        tn = lambda t: None if t is None else t.numpy()  # noqa: E731
        return list(linear_oracle.corr_moments_affine(fixed.numpy(), moving.numpy(), Af, bf, Am, bm, vsize, stride, tn(fixed_mask),
                                                      tn(moving_mask), jitter=jit(), moving_gradient=mgrad()))

    def fake_values(metric, fixed, fsize, moving, msize, Af, bf, Ams, bms, vsize, stride, fixed_mask=None, moving_mask=None):
        out = np.zeros((len(Ams), 6))
        for c, (Am, bm) in enumerate(zip(Ams, bms)):
            r = (fake_meansq if metric == 0 else fake_corr)(fixed, fsize, moving, msize, Af, bf, np.asarray(Am).ravel(), bm, vsize, stride,
                                                            fixed_mask, moving_mask)
            out[c, :2 if metric == 0 else 6] = r[:2 if metric == 0 else 6]
        return out

    from platipy_amd.registration import linear as _linear

    sa(_linear, "NATIVE_OPTIMISER", False)        # the native loop would run every probe through the emulated kernels
    sa(_linear, "LINE_SEARCH_SPECULATION", 1)     # sequential probes: the numpy stand-in gains nothing from batching
    sa(be.ctx, "metric_values_affine", fake_values)
    sa(be.ctx, "meansq_affine", fake_meansq)
    sa(be.ctx, "corr_moments_affine", fake_corr)
    return be


def sphere_case(i, shape=(30, 64, 64)):
    """The reference's synthetic cardiac case i (platipy/imaging/tests/test_cardiac.py:43-71) at half size:
    a -1000 volume holding a sphere of value 1, its mask, a small sub-structure, slightly different spacing."""
    nz, ny, nx = shape
    zz, yy, xx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    m = (zz - (nz // 2 + i)) ** 2 + (yy - (ny // 2 + i)) ** 2 + (xx - nx // 2) ** 2 <= (nx * 25 // 128) ** 2
    sub = (zz - (nz // 2 + i)) ** 2 + (yy - (ny // 2 - 2 + i)) ** 2 + (xx - (nx // 2 - 2)) ** 2 <= (nx * 5 // 128 + 1) ** 2
    ct = np.where(m, 1.0, -1000.0).astype(np.float32)
    return ct, m.astype(np.uint8), sub.astype(np.uint8), (0.9 + i * 0.01, 0.9 + i * 0.01, 2.5 + i * 0.01)


def record_stats(name, stats):
    """Keep a test's MEASURED error statistics: one JSON per test under
    $PP_STATS_DIR (default gpurun_out/parity_stats/, which gpurun merges back); the builder commits the collected
    file under profiles/.  Never fails a test."""
    import json

    d = os.environ.get("PP_STATS_DIR", os.path.join(ROOT, "gpurun_out", "parity_stats"))
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".json"), "w") as fh:
            json.dump(stats, fh, indent=1, sort_keys=True, default=float)
    except OSError:
        pass
