import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU test")


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, plus: the library reads its PP_* switches from the environment ONCE (pp_env), so setting or
    deleting one here re-takes the snapshot (pp_reload_switches), and so does the undo at the end of the test."""
    from platipy_amd import _lib

    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def setenv_(name, value, prepend=None):
        setenv(name, value, prepend)
        if name.startswith("PP_"):
            _lib.reload_switches()

    def delenv_(name, raising=True):
        delenv(name, raising)
        if name.startswith("PP_"):
            _lib.reload_switches()

    monkeypatch.setenv, monkeypatch.delenv = setenv_, delenv_
    yield monkeypatch
    monkeypatch.undo()
    _lib.reload_switches()


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
    import platipy_amd

    if request.param == "emu":
        from tests.helpers import install_emu_runtime

        install_emu_runtime(lambda obj, name, value: monkeypatch.setattr(obj, name, value, raising=False))
    else:
        request.getfixturevalue("gpu_backend")
    return platipy_amd
