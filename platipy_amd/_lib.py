"""ctypes binding of the C ABI in include/platipy_amd.h (libplatipy_hip.so, gfx950).

The product path has no CPU fallback: if the shared library is missing or fails to load, every
operation raises.  `Context` methods take raw device addresses (ints); `ptr()` extracts one from
a torch tensor.  The same binding class is pointed at the CPU-emulated build of the identical
kernel sources by the test-suite only (tests/emu), never by the package itself.
"""
import ctypes as C

import numpy as np
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libplatipy_hip.so")

PP_OK = 0
INTERP_NEAREST = 1
INTERP_LINEAR = 2
INTERP_BSPLINE = 3
DEMONS_AUTO, DEMONS_STAGED, DEMONS_FUSED = 0, 1, 2
ABI_VERSION = 2
HISTORY_CAPACITY = 4096     # PP_DEMONS_HISTORY_CAPACITY: iterations of one Execute whose metric / RMS change the device ring keeps


class Geom(C.Structure):
    _fields_ = [("size", C.c_int * 3), ("spacing", C.c_double * 3), ("origin", C.c_double * 3),
                ("direction", C.c_double * 9)]


class MiBins(C.Structure):
    """pp_mi_bins"""
    _fields_ = [("nbins", C.c_int), ("kernel", C.c_int), ("f_bin", C.c_double), ("f_norm_min", C.c_double), ("m_bin", C.c_double),
                ("m_norm_min", C.c_double)]


MI_MATTES, MI_JOINT = 0, 1


class LinregLevel(C.Structure):     # pp_linreg_level
    _fields_ = [("model", C.c_int), ("metric", C.c_int), ("optimizer", C.c_int), ("iterations", C.c_int), ("vsize", C.c_int * 3),
                ("stride", C.c_int), ("speculation", C.c_int), ("flags", C.c_int),
                ("v_i2p", C.c_double * 9), ("v_origin", C.c_double * 3), ("f_p2i", C.c_double * 9), ("f_origin", C.c_double * 3),
                ("m_p2i", C.c_double * 9), ("m_origin", C.c_double * 3), ("init_matrix", C.c_double * 9),
                ("init_offset", C.c_double * 3), ("center", C.c_double * 3), ("v_min_spacing", C.c_double)]


class LinregStats(C.Structure):     # pp_linreg_stats
    _fields_ = [("iterations", C.c_int), ("evaluations", C.c_int), ("stop", C.c_int), ("reserved", C.c_int), ("value", C.c_double),
                ("learning_rate", C.c_double)]


ERR_NO_OVERLAP = -6
MODEL_TRANSLATION, MODEL_VERSOR_RIGID, MODEL_SIMILARITY, MODEL_SCALE, MODEL_AFFINE, MODEL_EULER, MODEL_SCALE_VERSOR, MODEL_SCALE_SKEW_VERSOR = range(8)
OPT_GD, OPT_GD_LINE_SEARCH = 0, 1
LINREG_RETURN_BEST = 1


class DemonsParams(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("sigma_d_vox", C.c_double * 3),
        ("sigma_u_vox", C.c_double * 3),
        ("smooth_displacement", C.c_int),
        ("smooth_update", C.c_int),
        ("max_rms_error", C.c_double),
        ("max_step_length", C.c_double),
        ("intensity_threshold", C.c_double),
        ("denominator_threshold", C.c_double),
        ("max_error", C.c_double),
        ("max_kernel_width", C.c_int),
        ("variant", C.c_int),
    ]


class DemonsStats(C.Structure):
    _fields_ = [
        ("metric", C.c_double),
        ("rms_change", C.c_double),
        ("sum_sq_diff", C.c_double),
        ("sum_sq_change", C.c_double),
        ("n_pixels", C.c_int64),
        ("elapsed_iterations", C.c_int),
        ("halted", C.c_int),
    ]


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int), ("total_ms", C.c_double)]


class PlatipyAmdError(RuntimeError):
    pass


_P = C.c_void_p
_SIGNATURES = {
    "pp_abi_version": (C.c_int, []),
    "pp_create": (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    "pp_destroy": (None, [_P]),
    "pp_last_error": (C.c_char_p, [_P]),
    "pp_set_stream": (C.c_int, [_P, _P]),
    "pp_sync": (C.c_int, [_P]),
    "pp_workspace_bytes": (C.c_size_t, [_P]),
    "pp_profile_enable": (C.c_int, [_P, C.c_int]),
    "pp_profile_read": (C.c_int, [_P, C.POINTER(ProfileEntry), C.c_int]),
    "pp_gauss_taps": (C.c_int, [C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double), C.c_int]),
    "pp_demons_default_params": (None, [C.POINTER(DemonsParams)]),
    "pp_discrete_gaussian_f32": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.c_double, C.c_int, C.c_int]),
    "pp_discrete_gaussian_rows_f32": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                C.c_double, C.c_int, C.c_int, _P, _P]),
    "pp_smooth_field_f32": (C.c_int, [_P, _P, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_double, C.c_int]),
    "pp_recursive_gaussian_field_f32": (C.c_int, [_P, _P, C.POINTER(Geom), C.POINTER(C.c_double)]),
    "pp_recursive_gaussian_f32": (C.c_int, [_P, _P, _P, C.POINTER(Geom), C.POINTER(C.c_double)]),
    "pp_warp_f32": (C.c_int, [_P, _P, _P, C.POINTER(Geom), C.c_float, _P]),
    "pp_resample_f32": (C.c_int, [_P, _P, C.POINTER(Geom), C.POINTER(Geom), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  _P, C.c_int, C.c_double, _P]),
    "pp_resample_u8": (C.c_int, [_P, _P, C.POINTER(Geom), C.POINTER(Geom), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 _P, C.c_int, C.c_double, _P]),
    "pp_resample_field_f32": (C.c_int, [_P, _P, C.POINTER(Geom), C.POINTER(Geom), _P]),
    "pp_bspline_prefilter_f32": (C.c_int, [_P, _P, C.POINTER(C.c_int), _P]),
    "pp_compose_field_f32": (C.c_int, [_P, _P, _P, C.POINTER(Geom)]),
    "pp_transform_to_field_f32": (C.c_int, [_P, C.POINTER(Geom), C.POINTER(C.c_double), C.POINTER(C.c_double), _P, _P]),
    "pp_demons_force_f32": (C.c_int, [_P, _P, _P, C.POINTER(Geom), C.POINTER(DemonsParams), _P, C.POINTER(DemonsStats)]),
    "pp_demons_execute_f32": (C.c_int, [_P, _P, _P, C.POINTER(Geom), C.POINTER(DemonsParams), _P, C.POINTER(DemonsStats)]),
    "pp_demons_history": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
    "pp_weight_map_local_f32": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_double, C.c_double, _P]),
    "pp_weight_map_block_f32": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_double, C.c_double, _P]),
    "pp_sum_sq_diff_f32": (C.c_int, [_P, _P, _P, C.c_size_t, C.POINTER(C.c_double)]),
    "pp_fuse_accumulate_u8": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t]),
    "pp_fuse_accumulate_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t]),
    "pp_fuse_divide_f32": (C.c_int, [_P, _P, _P, _P, C.c_size_t]),
    "pp_minmax_f32": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "pp_rescale_threshold_f32": (C.c_int, [_P, _P, C.c_size_t, C.c_float, C.c_float, C.c_float]),
    "pp_binary_threshold_f32": (C.c_int, [_P, _P, C.c_size_t, C.c_double, C.c_double, _P]),
    "pp_fillhole_largest_component_u8": (C.c_int, [_P, _P, C.POINTER(C.c_int), C.c_int, _P, C.POINTER(C.c_int64)]),
    "pp_binary_morph_ball_u8": (C.c_int, [_P, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, _P]),
    "pp_bounding_box": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pp_label_contour_u8": (C.c_int, [_P, _P, C.POINTER(C.c_int), _P]),
    "pp_distance_map_f32": (C.c_int, [_P, _P, C.POINTER(Geom), C.c_int, C.c_int, _P]),
    "pp_meansq_affine_f32": (C.c_int, [_P, _P, C.POINTER(C.c_int), _P, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.POINTER(C.c_int), C.c_int, _P, _P, C.POINTER(C.c_double)]),
    "pp_corr_moments_affine_f32": (C.c_int, [_P, _P, C.POINTER(C.c_int), _P, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                             C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                             C.POINTER(C.c_int), C.c_int, _P, _P, C.POINTER(C.c_double)]),
    "pp_metric_values_affine_f32": (C.c_int, [_P, C.c_int, _P, C.POINTER(C.c_int), _P, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                              C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                              C.POINTER(C.c_int), C.c_int, _P, _P, C.POINTER(C.c_double)]),
    "pp_linear_set_sample_jitter": (C.c_int, [_P, _P, C.c_size_t]),
    "pp_linear_set_moving_gradient": (C.c_int, [_P, _P, C.POINTER(C.c_int)]),
    "pp_linear_set_moving_gradient_packed": (C.c_int, [_P, _P]),
    "pp_recursive_gaussian_pass_f32": (C.c_int, [_P, _P, _P, C.POINTER(Geom), C.c_int, C.c_double, C.c_int, C.c_int]),
    "pp_mi_histogram_f32": (C.c_int, [_P, _P, C.POINTER(C.c_int), _P, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, _P, _P, C.POINTER(MiBins),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pp_mi_gradient_f32": (C.c_int, [_P, _P, C.POINTER(C.c_int), _P, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, _P, _P, C.POINTER(MiBins),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pp_linear_num_parameters": (C.c_int, [C.c_int]),
    "pp_reload_switches": (None, []),
    "pp_linear_optimize_f32": (C.c_int, [_P, _P, C.POINTER(C.c_int), _P, C.POINTER(C.c_int), _P, _P, C.POINTER(LinregLevel),
                                         C.POINTER(C.c_double), C.POINTER(LinregStats), C.POINTER(C.c_double), C.c_int]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load(path=None):
    """dlopen the C-ABI library and declare every entry point.  Raises if it is absent."""
    path = path or DEFAULT_LIB
    if not os.path.exists(path):
        raise PlatipyAmdError(
            f"{path} not found: build it with `python -m platipy_amd._build` (needs hipcc). "
            "platipy_amd has no CPU fallback.")
    try:
        dll = C.CDLL(path)
    except OSError as e:  # e.g. libamdhip64 missing
        raise PlatipyAmdError(f"cannot load {path}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(dll, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if dll.pp_abi_version() != ABI_VERSION:
        raise PlatipyAmdError(f"{path}: ABI version {dll.pp_abi_version()} != {ABI_VERSION}")
    _LOADED.append(dll)
    return dll


_LOADED = []


def reload_switches():
    """pp_reload_switches() on every library this process loaded: the PP_* measurement switches are read from the environment
    once; tests and A/B tools that flip one inside a process call this afterwards (not while kernels are being launched from
    another thread)."""
    for d in _LOADED:
        d.pp_reload_switches()


_DLL = None


def dll():
    global _DLL
    if _DLL is None:
        _DLL = load()
    return _DLL


def ptr(x):
    """Device (or, in the emulated test build, host) address of a tensor/array, or None."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return x.ctypes.data  # numpy (test builds only)


def make_geom(size, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=(1, 0, 0, 0, 1, 0, 0, 0, 1)):
    g = Geom()
    g.size[:] = [int(s) for s in size]
    g.spacing[:] = [float(s) for s in spacing]
    g.origin[:] = [float(s) for s in origin]
    g.direction[:] = [float(s) for s in direction]
    return g


def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _dn(v, n):
    if v is None:
        return None
    return (C.c_double * n)(*[float(x) for x in v])


def gauss_taps(variance, max_error, max_kernel_width, lib=None):
    lib = lib or dll()
    buf = (C.c_double * 1024)()
    r = lib.pp_gauss_taps(float(variance), float(max_error), int(max_kernel_width), buf, 1024)
    if r < 0:
        raise PlatipyAmdError(f"pp_gauss_taps failed ({r})")
    return [buf[i] for i in range(2 * r + 1)]


class Context:
    """One pp_ctx: a (device, stream) pair plus its scratch memory."""

    def __init__(self, device=0, stream=0, lib=None):
        self.lib = lib or dll()
        h = _P()
        rc = self.lib.pp_create(int(device), _P(stream or None), C.byref(h))
        if rc != PP_OK:
            raise PlatipyAmdError(f"pp_create(device={device}) failed ({rc})")
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            self.lib.pp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != PP_OK:
            msg = self.lib.pp_last_error(self.h)
            raise PlatipyAmdError(f"{what} failed ({rc}): {msg.decode(errors='replace') if msg else ''}")

    def set_stream(self, stream):
        self._chk(self.lib.pp_set_stream(self.h, _P(stream or None)), "pp_set_stream")

    def sync(self):
        self._chk(self.lib.pp_sync(self.h), "pp_sync")

    def workspace_bytes(self):
        return self.lib.pp_workspace_bytes(self.h)

    def profile_enable(self, on=True, every=1):
        """Per-launch HIP events around the demons kernels; `every` = k > 1 brackets every k-th launch of each kernel."""
        self._chk(self.lib.pp_profile_enable(self.h, (max(1, int(every)) if on else 0)), "pp_profile_enable")

    def profile_read(self):
        """{kernel name: (launches, total ms)} since the last read (synchronises the stream)."""
        buf = (ProfileEntry * 32)()
        n = self.lib.pp_profile_read(self.h, buf, 32)
        if n < 0:
            self._chk(n, "pp_profile_read")
        return {buf[i].name.decode(): (buf[i].launches, buf[i].total_ms) for i in range(n)}

    def default_demons_params(self):
        p = DemonsParams()
        self.lib.pp_demons_default_params(C.byref(p))
        return p

    # -- FIR / IIR ------------------------------------------------------------------
    def discrete_gaussian(self, src, dst, size, spacing, variance, max_error=0.01, max_kernel_width=32, use_spacing=True):
        self._chk(self.lib.pp_discrete_gaussian_f32(self.h, ptr(src), ptr(dst), _i3(size), _d3(spacing), _d3(variance),
                                                    float(max_error), int(max_kernel_width), int(bool(use_spacing))),
                  "pp_discrete_gaussian_f32")

    def discrete_gaussian_rows(self, src, dst, size, spacing, variance, need_y, need_z, max_error=0.01, max_kernel_width=32,
                               use_spacing=True):
        """DiscreteGaussian valid only at rows (y, z) with need_y[y] and need_z[z] (uint8 device masks)."""
        self._chk(self.lib.pp_discrete_gaussian_rows_f32(self.h, ptr(src), ptr(dst), _i3(size), _d3(spacing), _d3(variance),
                                                         float(max_error), int(max_kernel_width), int(bool(use_spacing)), ptr(need_y),
                                                         ptr(need_z)), "pp_discrete_gaussian_rows_f32")

    def smooth_field(self, field, size, sigma_vox, max_error=0.1, max_kernel_width=30):
        self._chk(self.lib.pp_smooth_field_f32(self.h, ptr(field), _i3(size), _d3(sigma_vox), float(max_error),
                                               int(max_kernel_width)), "pp_smooth_field_f32")

    def recursive_gaussian_field(self, field, geom, sigma):
        self._chk(self.lib.pp_recursive_gaussian_field_f32(self.h, ptr(field), C.byref(geom), _d3(sigma)),
                  "pp_recursive_gaussian_field_f32")

    def recursive_gaussian(self, src, dst, geom, sigma):
        self._chk(self.lib.pp_recursive_gaussian_f32(self.h, ptr(src), ptr(dst), C.byref(geom), _d3(sigma)),
                  "pp_recursive_gaussian_f32")

    # -- warp / resample ------------------------------------------------------------
    def warp(self, moving, field, geom, edge_value, out):
        self._chk(self.lib.pp_warp_f32(self.h, ptr(moving), ptr(field), C.byref(geom), float(edge_value), ptr(out)),
                  "pp_warp_f32")

    def resample(self, src, gin, gout, out, affine_A=None, affine_t=None, field=None, interp=INTERP_LINEAR,
                 default_value=0.0, u8=False):
        fn = self.lib.pp_resample_u8 if u8 else self.lib.pp_resample_f32
        self._chk(fn(self.h, ptr(src), C.byref(gin), C.byref(gout), _dn(affine_A, 9), _dn(affine_t, 3), ptr(field),
                     int(interp), float(default_value), ptr(out)), "pp_resample")

    def bspline_prefilter(self, src, size, out):
        """B-spline (order 3) coefficients of a fp32 volume (`out` may be `src`)."""
        self._chk(self.lib.pp_bspline_prefilter_f32(self.h, ptr(src), _i3(size), ptr(out)), "pp_bspline_prefilter_f32")

    def resample_field(self, src, gin, gout, out):
        self._chk(self.lib.pp_resample_field_f32(self.h, ptr(src), C.byref(gin), C.byref(gout), ptr(out)),
                  "pp_resample_field_f32")

    def transform_to_field(self, geom, A, t, add_field, out):
        """out = (A - I) p + t (+ add_field) on the grid `geom` (sitk.TransformToDisplacementField for a linear map)."""
        a = (C.c_double * 9)(*[float(v) for v in np.asarray(A, dtype=np.float64).ravel()])
        tt = (C.c_double * 3)(*[float(v) for v in np.asarray(t, dtype=np.float64).ravel()])
        self._chk(self.lib.pp_transform_to_field_f32(self.h, C.byref(geom), a, tt, ptr(add_field) if add_field is not None else None,
                                                     ptr(out)), "pp_transform_to_field_f32")

    def compose_field(self, total, it, geom):
        self._chk(self.lib.pp_compose_field_f32(self.h, ptr(total), ptr(it), C.byref(geom)), "pp_compose_field_f32")

    # -- demons ---------------------------------------------------------------------
    def demons_force(self, fixed, warped, geom, params, update, want_stats=True):
        st = DemonsStats()
        self._chk(self.lib.pp_demons_force_f32(self.h, ptr(fixed), ptr(warped), C.byref(geom), C.byref(params), ptr(update),
                                               C.byref(st) if want_stats else None), "pp_demons_force_f32")
        return st if want_stats else None

    def demons_execute(self, fixed, moving, geom, params, field, want_stats=True):
        st = DemonsStats()
        self._chk(self.lib.pp_demons_execute_f32(self.h, ptr(fixed), ptr(moving), C.byref(geom), C.byref(params), ptr(field),
                                                 C.byref(st) if want_stats else None), "pp_demons_execute_f32")
        return st if want_stats else None

    def demons_history(self, cap=4096):
        """-> [(metric, rms_change)] per iteration of the last demons_execute on this context (an sitkIterationEvent
        observer's view of the filter, deformable.py:260-264)."""
        m, r = (C.c_double * cap)(), (C.c_double * cap)()
        n = self.lib.pp_demons_history(self.h, m, r, cap)
        self._chk(n if n < 0 else 0, "pp_demons_history")
        kept = min(n, cap, HISTORY_CAPACITY)
        if n > kept:
            import warnings

            warnings.warn(f"demons_history: {n} iterations ran, the device ring keeps the first {HISTORY_CAPACITY}; "
                          f"{n - kept} later iterations have no recorded metric / RMS change", RuntimeWarning)
        self.last_history_iterations = n     # iterations that ran (may exceed the number of entries)
        return [(m[k], r[k]) for k in range(kept)]

    # -- fusion ---------------------------------------------------------------------
    def weight_map_local(self, target, moving, size, spacing, sigma, epsilon, out):
        self._chk(self.lib.pp_weight_map_local_f32(self.h, ptr(target), ptr(moving), _i3(size), _d3(spacing), float(sigma),
                                                   float(epsilon), ptr(out)), "pp_weight_map_local_f32")

    def weight_map_block(self, target, moving, size, radius, factor, gain, weight):
        self._chk(self.lib.pp_weight_map_block_f32(self.h, ptr(target), ptr(moving), _i3(size), _i3(radius), float(factor), float(gain),
                                                   ptr(weight)), "pp_weight_map_block_f32")

    def sum_sq_diff(self, a, b, n):
        r = C.c_double()
        self._chk(self.lib.pp_sum_sq_diff_f32(self.h, ptr(a), ptr(b), int(n), C.byref(r)), "pp_sum_sq_diff_f32")
        return r.value

    def fuse_accumulate(self, weight, label, wsum, wlsum, n):
        """wsum += w; wlsum += w * label.  `label` is uint8 (masks) or float32 (probabilistic labels)."""
        if getattr(label, "dtype", None) is not None and str(label.dtype).endswith("float32"):
            self._chk(self.lib.pp_fuse_accumulate_f32(self.h, ptr(weight), ptr(label), ptr(wsum), ptr(wlsum), int(n)),
                      "pp_fuse_accumulate_f32")
        else:
            self._chk(self.lib.pp_fuse_accumulate_u8(self.h, ptr(weight), ptr(label), ptr(wsum), ptr(wlsum), int(n)),
                      "pp_fuse_accumulate_u8")

    def fuse_divide(self, wlsum, wsum, out, n):
        self._chk(self.lib.pp_fuse_divide_f32(self.h, ptr(wlsum), ptr(wsum), ptr(out), int(n)), "pp_fuse_divide_f32")

    def minmax(self, src, n):
        lo, hi = C.c_float(), C.c_float()
        self._chk(self.lib.pp_minmax_f32(self.h, ptr(src), int(n), C.byref(lo), C.byref(hi)), "pp_minmax_f32")
        return lo.value, hi.value

    def rescale_threshold(self, data, n, in_min, in_max, lower):
        self._chk(self.lib.pp_rescale_threshold_f32(self.h, ptr(data), int(n), float(in_min), float(in_max), float(lower)),
                  "pp_rescale_threshold_f32")

    def binary_threshold(self, prob, n, max_value, threshold, out):
        self._chk(self.lib.pp_binary_threshold_f32(self.h, ptr(prob), int(n), float(max_value), float(threshold), ptr(out)),
                  "pp_binary_threshold_f32")

    def fillhole_largest_component(self, mask, size, out, fill_holes=True, want_count=False):
        n = C.c_int64(0)
        self._chk(self.lib.pp_fillhole_largest_component_u8(self.h, ptr(mask), _i3(size), int(bool(fill_holes)), ptr(out),
                                                            C.byref(n) if want_count else None), "pp_fillhole_largest_component_u8")
        return n.value if want_count else None

    def binary_morph_ball(self, mask, size, radius, op, out):
        """op: 0 dilate, 1 erode, 2 closing (safe border); ITK ball of `radius` voxels (x, y, z)."""
        self._chk(self.lib.pp_binary_morph_ball_u8(self.h, ptr(mask), _i3(size), _i3(radius), int(op), ptr(out)),
                  "pp_binary_morph_ball_u8")

    def bounding_box(self, data, size, is_float):
        """-> [xmin, xmax, ymin, ymax, zmin, zmax] of the voxels > 0 (xmin > xmax when there are none)."""
        box = (C.c_int * 6)()
        self._chk(self.lib.pp_bounding_box(self.h, ptr(data), 1 if is_float else 0, _i3(size), box), "pp_bounding_box")
        return [box[i] for i in range(6)]

    def label_contour(self, mask, size, out):
        self._chk(self.lib.pp_label_contour_u8(self.h, ptr(mask), _i3(size), ptr(out)), "pp_label_contour_u8")

    def distance_map(self, mask, geom, out, signed=False, inside_positive=False):
        self._chk(self.lib.pp_distance_map_f32(self.h, ptr(mask), C.byref(geom), int(bool(signed)), int(bool(inside_positive)),
                                               ptr(out)), "pp_distance_map_f32")

    def meansq_affine(self, fixed, fsize, moving, msize, Af, bf, Am, bm, vsize, stride, fixed_mask=None, moving_mask=None):
        """-> (sum sq diff, count, dAm[9], dbm[3]) as a list of 14 floats."""
        res = (C.c_double * 14)()
        self._chk(self.lib.pp_meansq_affine_f32(self.h, ptr(fixed), _i3(fsize), ptr(moving), _i3(msize), _dn(Af, 9), _dn(bf, 3),
                                                _dn(Am, 9), _dn(bm, 3), _i3(vsize), int(stride), ptr(fixed_mask),
                                                ptr(moving_mask), res), "pp_meansq_affine_f32")
        return [res[i] for i in range(14)]

    def metric_values_affine(self, metric, fixed, fsize, moving, msize, Af, bf, Ams, bms, vsize, stride, fixed_mask=None,
                             moving_mask=None):
        """Values only for len(Ams) <= 16 candidate maps in one launch -> array [ncand, 6] (pp_metric_values_affine_f32)."""
        import numpy as np

        k = len(Ams)
        am = np.ascontiguousarray(np.asarray(Ams, dtype=np.float64).reshape(k, 9))
        bm = np.ascontiguousarray(np.asarray(bms, dtype=np.float64).reshape(k, 3))
        res = np.zeros((k, 6), dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self._chk(self.lib.pp_metric_values_affine_f32(self.h, int(metric), ptr(fixed), _i3(fsize), ptr(moving), _i3(msize), _dn(Af, 9),
                                                       _dn(bf, 3), k, am.ctypes.data_as(dp), bm.ctypes.data_as(dp), _i3(vsize),
                                                       int(stride), ptr(fixed_mask), ptr(moving_mask), res.ctypes.data_as(dp)),
                  "pp_metric_values_affine_f32")
        return res

    def set_sample_jitter(self, jitter):
        """ITK's per-sample jitter for the metric entry points (pp_linear_set_sample_jitter): a contiguous float32 device
        tensor [nsamples, 3] in virtual-index units, kept alive by this context until replaced; None restores the lattice."""
        if jitter is None:
            self._chk(self.lib.pp_linear_set_sample_jitter(self.h, None, 0), "pp_linear_set_sample_jitter")
        else:
            shape, dtype = tuple(jitter.shape), str(jitter.dtype).replace("torch.", "")
            contiguous = jitter.is_contiguous() if hasattr(jitter, "is_contiguous") else jitter.flags["C_CONTIGUOUS"]
            if len(shape) != 2 or shape[1] != 3 or dtype != "float32" or not contiguous:
                raise ValueError("set_sample_jitter: a contiguous float32 array [nsamples, 3]")
            self._chk(self.lib.pp_linear_set_sample_jitter(self.h, ptr(jitter), int(jitter.shape[0])), "pp_linear_set_sample_jitter")
        self._sample_jitter = jitter

    def set_moving_gradient(self, gradient, msize=None, packed=None):
        """ITK's filtered gradient image for the gradient-bearing metric entry points (pp_linear_set_moving_gradient): a
        contiguous float32 device tensor [3, Z, Y, X] in moving-index units, kept alive by this context until replaced; None
        restores the interpolant's analytic gradient.  `packed`: optionally the same image with the intensity, float32
        [Z, Y, X, 4] = (gx, gy, gz, m) (pp_linear_set_moving_gradient_packed)."""
        self._moving_gradient_packed = None
        if gradient is None:
            self._chk(self.lib.pp_linear_set_moving_gradient(self.h, None, None), "pp_linear_set_moving_gradient")
        else:
            shape = tuple(gradient.shape)
            if len(shape) != 4 or shape[0] != 3 or str(gradient.dtype).replace("torch.", "") != "float32":
                raise ValueError("set_moving_gradient: a float32 array [3, Z, Y, X]")
            size = msize if msize is not None else (shape[3], shape[2], shape[1])
            self._chk(self.lib.pp_linear_set_moving_gradient(self.h, ptr(gradient), _i3(size)), "pp_linear_set_moving_gradient")
            if packed is not None:
                if tuple(packed.shape) != shape[1:] + (4,) or str(packed.dtype).replace("torch.", "") != "float32":
                    raise ValueError("set_moving_gradient: `packed` must be a float32 array [Z, Y, X, 4]")
                self._chk(self.lib.pp_linear_set_moving_gradient_packed(self.h, ptr(packed)), "pp_linear_set_moving_gradient_packed")
                self._moving_gradient_packed = packed
        self._moving_gradient = gradient

    def recursive_gaussian_pass(self, src, dst, geom, axis, sigma, order=0, normalize_across_scale=False):
        """One directional pass of itk::RecursiveGaussianImageFilter (pp_recursive_gaussian_pass_f32): order 0 / 1 along `axis`."""
        self._chk(self.lib.pp_recursive_gaussian_pass_f32(self.h, ptr(src), ptr(dst), C.byref(geom), int(axis), float(sigma), int(order),
                                                          int(bool(normalize_across_scale))), "pp_recursive_gaussian_pass_f32")

    def linear_optimize(self, fixed, fsize, moving, msize, level, params, fixed_mask=None, moving_mask=None, history=0):
        """One level of linear_registration's optimisation in the library (pp_linear_optimize_f32).
        -> (params list, LinregStats, history list).  Raises PlatipyAmdError; .code == ERR_NO_OVERLAP when the images
        do not overlap at the start."""
        n = self.lib.pp_linear_num_parameters(int(level.model))
        p = (C.c_double * n)(*[float(v) for v in params])
        st = LinregStats()
        hist = (C.c_double * max(1, int(history)))()
        rc = self.lib.pp_linear_optimize_f32(self.h, ptr(fixed), _i3(fsize), ptr(moving), _i3(msize), ptr(fixed_mask), ptr(moving_mask),
                                             C.byref(level), p, C.byref(st), hist, int(history))
        if rc:
            msg = self.lib.pp_last_error(self.h)
            err = PlatipyAmdError(f"pp_linear_optimize_f32 failed ({rc}): {msg.decode(errors='replace') if msg else ''}")
            err.code = rc
            raise err
        return [p[i] for i in range(n)], st, [hist[i] for i in range(min(int(history), st.iterations))]

    def mi_histogram(self, fixed, fsize, moving, msize, Af, bf, Am, bm, vsize, stride, bins, fixed_mask=None, moving_mask=None):
        """Joint intensity histogram of the valid sample pairs -> (hist [nbins, nbins] float64, row = fixed bin; count)."""
        hist = np.zeros((bins.nbins, bins.nbins), dtype=np.float64)
        count = C.c_double()
        self._chk(self.lib.pp_mi_histogram_f32(self.h, ptr(fixed), _i3(fsize), ptr(moving), _i3(msize), _dn(Af, 9), _dn(bf, 3), _dn(Am, 9),
                                               _dn(bm, 3), _i3(vsize), int(stride), ptr(fixed_mask), ptr(moving_mask), C.byref(bins),
                                               hist.ctypes.data_as(C.POINTER(C.c_double)), C.byref(count)), "pp_mi_histogram_f32")
        return hist, count.value

    def mi_gradient(self, fixed, fsize, moving, msize, Af, bf, Am, bm, vsize, stride, bins, table, fixed_mask=None, moving_mask=None):
        """sum_s w_s g_s (v_q | 1) with w_s from the score `table` [nbins, nbins] -> 12 floats (d/dAm row-major 9, d/dbm 3)."""
        tab = np.ascontiguousarray(table, dtype=np.float64)
        res = (C.c_double * 12)()
        self._chk(self.lib.pp_mi_gradient_f32(self.h, ptr(fixed), _i3(fsize), ptr(moving), _i3(msize), _dn(Af, 9), _dn(bf, 3), _dn(Am, 9),
                                              _dn(bm, 3), _i3(vsize), int(stride), ptr(fixed_mask), ptr(moving_mask), C.byref(bins),
                                              tab.ctypes.data_as(C.POINTER(C.c_double)), res), "pp_mi_gradient_f32")
        return np.array([res[i] for i in range(12)])

    def corr_moments_affine(self, fixed, fsize, moving, msize, Af, bf, Am, bm, vsize, stride, fixed_mask=None, moving_mask=None):
        """-> the 42 raw moments of pp_corr_moments_affine_f32."""
        res = (C.c_double * 42)()
        self._chk(self.lib.pp_corr_moments_affine_f32(self.h, ptr(fixed), _i3(fsize), ptr(moving), _i3(msize), _dn(Af, 9), _dn(bf, 3),
                                                      _dn(Am, 9), _dn(bm, 3), _i3(vsize), int(stride), ptr(fixed_mask),
                                                      ptr(moving_mask), res), "pp_corr_moments_affine_f32")
        return [res[i] for i in range(42)]
