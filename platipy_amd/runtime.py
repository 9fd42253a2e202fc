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


# --------------------------------------------------------------------------------------
# Staggered atlas chains (projects/multiatlas.py::_map_atlases).
#
# Several atlas chains share one GPU on several HIP streams.  Their phases differ in kind: linear registration and the coarse
# demons levels are latency-bound (small kernels, host round trips), the finest demons level and the full-resolution
# resamples behind it are throughput-bound (every CU busy, HBM-bound).  Left alone, chains started together stay in lockstep:
# four finest levels share the chip and run at 41 Gvoxel/s in aggregate where one alone runs at 50, and the latency-bound
# phases only ever overlap each other (profiles/round5_streams_timeline_8queues.md).  The turnstile lets ONE chain at a time
# through its throughput-bound phase -- on the device, by an event chain between the streams, not by blocking a host thread for
# the phase's duration -- so the other chains' latency-bound phases run underneath it.


class Turnstile:
    """One stream at a time through a section of device work.  `section(stream)`: the work enqueued inside starts after the
    previous section's work has finished on the device, whichever stream that was on.  The host lock is held only while the
    section's launches are enqueued (an event must have been recorded before another stream can be told to wait for it)."""

    def __init__(self):
        self._lock = threading.Lock()
        self._last = None
        self.sections = 0

    def section(self, stream):
        return _TurnstileSection(self, stream)


class _TurnstileSection:
    def __init__(self, turnstile, stream):
        self.t, self.stream = turnstile, stream

    def __enter__(self):
        self.t._lock.acquire()
        if self.t._last is not None:
            self.stream.wait_event(self.t._last)
        return self

    def __exit__(self, *exc):
        try:
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self.t._last = ev
            self.t.sections += 1
        finally:
            self.t._lock.release()
        return False


_SCHEDULE = threading.local()


class _NoSection:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def set_turnstile(turnstile, entry_slots=None):
    """Bind (or, with None, unbind) the calling THREAD to a turnstile: `exclusive()` sections entered by this thread then pass
    through it.  Called by the multi-atlas worker threads; nothing is bound outside them, so a single registration pays nothing.
    `entry_slots`: a threading.Semaphore bounding how many bound threads are inside `entry_slot()` at once (the chains' linear
    stage: admitting them one after the other is what staggers chains that were started together)."""
    _SCHEDULE.turnstile = turnstile
    _SCHEDULE.entry_slots = entry_slots


def entry_slot():
    """Context manager around a chain's first latency-bound stage; a no-op unless the thread was bound with `entry_slots`."""
    s = getattr(_SCHEDULE, "entry_slots", None)
    return s if s is not None else _NoSection()


def exclusive(device=None):
    """Context manager around a throughput-bound phase (registration/deformable.py: the demons levels above HEAVY_VOXELS and the
    full-resolution resamples behind them).  A no-op unless the calling thread was bound to a turnstile."""
    t = getattr(_SCHEDULE, "turnstile", None)
    if t is None:
        return _NoSection()
    idx = torch.device(device).index if device is not None else None
    return t.section(torch.cuda.current_stream(idx if idx is not None else torch.cuda.current_device()))


HEAVY_VOXELS = 4 << 20      # a demons level at or above this many voxels fills the chip by itself (kernels of >= 512 marching blocks)


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
    from .registration.linear import release_cached_jitter
    from .registration.utils import release_cached_masks

    release_cached_masks()
    release_cached_jitter()
