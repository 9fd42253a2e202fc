import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU test")


@pytest.fixture(scope="session")
def emu_backend():
    from tests.helpers import EmuBackend

    return EmuBackend()


@pytest.fixture(scope="session")
def gpu_backend():
    from tests.helpers import GpuBackend

    return GpuBackend()


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    """The HIP kernels behind the C ABI: on the GPU (real library, -m gpu) or -- CPU suite only --
    the same sources compiled against the test-only HIP stand-in (tests/emu)."""
    if request.param == "emu":
        return request.getfixturevalue("emu_backend")
    return request.getfixturevalue("gpu_backend")


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def host_api(request, monkeypatch):
    """The drop-in Python API (platipy_amd.*).  `gpu`: the product as shipped, tensors on cuda:0.
    `emu` (CPU suite only): the same host code with its context lookup pointed at the CPU-emulated
    kernels and tensors kept on the host -- test plumbing, not a product fallback."""
    import torch

    import platipy_amd
    from platipy_amd import runtime

    if request.param == "emu":
        be = request.getfixturevalue("emu_backend")
        monkeypatch.setattr(runtime, "context", lambda device=None: be.ctx)
        monkeypatch.setattr(runtime, "default_device", lambda: torch.device("cpu"))
        # linear_registration evaluates the metric thousands of times; emulating each 256-thread reduction
        # launch thread-by-thread is too slow for the CPU suite, so here (only) the metric evaluation is the
        # oracle's vectorised restatement.  The kernel itself is compared with that restatement, on the
        # emulator and on the GPU, in tests/test_linear.py::test_meansq_kernel_*.
        from oracle import linear_oracle

        def fake_meansq(fixed, fsize, moving, msize, Af, bf, Am, bm, vsize, stride, fixed_mask=None, moving_mask=None):
            tn = lambda t: None if t is None else t.numpy()  # noqa: E731
            return list(linear_oracle.meansq_affine(fixed.numpy(), moving.numpy(), Af, bf, Am, bm, vsize, stride, tn(fixed_mask),
                                                    tn(moving_mask)))

        monkeypatch.setattr(be.ctx, "meansq_affine", fake_meansq, raising=False)
    else:
        request.getfixturevalue("gpu_backend")
    return platipy_amd
