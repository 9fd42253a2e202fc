"""Per-device contexts of the C ABI, bound to torch's current stream."""
import threading

import torch

from . import _lib

_LOCK = threading.Lock()
_CTX = {}


def default_device():
    if not torch.cuda.is_available():
        raise _lib.PlatipyAmdError("platipy_amd needs a ROCm GPU: there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def context(device=None):
    """The pp_ctx bound to torch's CURRENT stream on `device` (default: current CUDA device).  One ctx -- and one
    scratch workspace -- per (device, stream): kernels order with surrounding torch work on that stream, and worker
    threads that each run under their own `torch.cuda.stream(...)` get independent contexts that persist across
    calls (streams are long-lived; nothing is keyed by thread)."""
    if not torch.cuda.is_available():
        raise _lib.PlatipyAmdError("platipy_amd needs a ROCm GPU: there is no CPU fallback")
    if device is None:
        idx = torch.cuda.current_device()
    else:
        idx = torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device()
    stream = torch.cuda.current_stream(idx).cuda_stream
    key = (idx, stream)
    with _LOCK:
        ctx = _CTX.get(key)
        if ctx is None:
            ctx = _lib.Context(idx, stream)
            _CTX[key] = ctx
    return ctx


def release_all():
    """Destroy every context (and its workspace) and drop the device-side caches of the Python layer."""
    import torch

    if torch.cuda.is_available():
        # nothing queued may still read what is about to be freed -- on ANY device that owns a context (synchronize() alone
        # waits for the current device only; ranks / threads may have driven others)
        with _LOCK:
            devices = {idx for idx, _ in _CTX}
        devices.add(torch.cuda.current_device())
        for idx in sorted(devices):
            torch.cuda.synchronize(idx)
    with _LOCK:
        for c in _CTX.values():
            c.close()
        _CTX.clear()
    from .registration.utils import release_cached_masks

    release_cached_masks()
