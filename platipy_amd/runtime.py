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
    """The pp_ctx for `device` (default: current CUDA device), re-pointed at torch's current stream so
    kernels order correctly with surrounding torch work.  One ctx per (thread, device)."""
    if not torch.cuda.is_available():
        raise _lib.PlatipyAmdError("platipy_amd needs a ROCm GPU: there is no CPU fallback")
    if device is None:
        idx = torch.cuda.current_device()
    else:
        idx = torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device()
    key = (threading.get_ident(), idx)
    stream = torch.cuda.current_stream(idx).cuda_stream
    with _LOCK:
        ctx = _CTX.get(key)
        if ctx is None:
            ctx = _lib.Context(idx, stream)
            _CTX[key] = ctx
            ctx._stream = stream
    if ctx._stream != stream:
        ctx.set_stream(stream)
        ctx._stream = stream
    return ctx


def release_all():
    with _LOCK:
        for c in _CTX.values():
            c.close()
        _CTX.clear()
