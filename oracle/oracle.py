"""oracle/oracle.py -- numpy/ctypes face of the CPU restatement.

TEST INFRASTRUCTURE ONLY (see oracle/pp_oracle.h).  PARITY UNPINNED: SimpleITK, which does
all of the reference's arithmetic on this path, is not available in the build image, and the
reference's tests hold no golden vectors for it.  Nothing under platipy_amd/ imports this.

The voxel-level arithmetic lives in pp_oracle.c; this file restates the reference's *Python*
orchestration around it, citing the reference file:line each function follows
(paths relative to /root/reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

INTERP_NEAREST = 1  # sitk.sitkNearestNeighbor
INTERP_LINEAR = 2   # sitk.sitkLinear


class _Geom(C.Structure):
    _fields_ = [
        ("size", C.c_int * 3),
        ("spacing", C.c_double * 3),
        ("origin", C.c_double * 3),
        ("direction", C.c_double * 9),
    ]


class DemonsStats(C.Structure):
    _fields_ = [
        ("metric", C.c_double),
        ("rms_change", C.c_double),
        ("sum_sq_diff", C.c_double),
        ("sum_sq_change", C.c_double),
        ("n_pixels", C.c_int64),
        ("elapsed_iterations", C.c_int),
    ]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "pp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_gaussian_operator.restype = C.c_int
        _LIB.orc_gaussian_operator.argtypes = [C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_int]
        _LIB.orc_num_threads.restype = C.c_int
    return _LIB


class Vol:
    """A volume + geometry.  arr is [Z,Y,X] (scalar) or [3,Z,Y,X] (planar vector field)."""

    def __init__(self, arr, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=None):
        self.arr = np.ascontiguousarray(arr)
        self.spacing = tuple(float(s) for s in spacing)
        self.origin = tuple(float(o) for o in origin)
        self.direction = tuple(float(d) for d in (direction or (1, 0, 0, 0, 1, 0, 0, 0, 1)))

    @property
    def size(self):
        s = self.arr.shape[-3:]
        return (s[2], s[1], s[0])

    def like(self, arr):
        return Vol(arr, self.spacing, self.origin, self.direction)

    def geom(self):
        g = _Geom()
        g.size[:] = self.size
        g.spacing[:] = self.spacing
        g.origin[:] = self.origin
        g.direction[:] = self.direction
        return g


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed rc={rc}")


# --------------------------------------------------------------------------------------
# thin wrappers


def gaussian_operator(variance, max_error, max_kernel_width):
    buf = np.zeros(8192, dtype=np.float64)
    r = lib().orc_gaussian_operator(float(variance), float(max_error), int(max_kernel_width), _p(buf), buf.size)
    if r < 0:
        raise RuntimeError("kernel too large")
    return buf[: 2 * r + 1].copy()


def discrete_gaussian(vol, variance, max_kernel_width=32, max_error=0.01, use_spacing=True):
    """sitk.DiscreteGaussian(image, variance, maximumKernelWidth=32, maximumError=0.01, useImageSpacing=True)."""
    a = np.ascontiguousarray(vol.arr, dtype=np.float32)
    out = np.empty_like(a)
    var = np.broadcast_to(np.asarray(variance, dtype=np.float64), (3,))
    _chk(
        lib().orc_discrete_gaussian_f32(
            _p(a), _p(out), _i3(vol.size), _d3(vol.spacing), _d3(var), C.c_double(max_error),
            C.c_int(int(max_kernel_width)), C.c_int(1 if use_spacing else 0)),
        "discrete_gaussian",
    )
    return vol.like(out)


def smooth_field(field, sigma_vox, max_error=0.1, max_kernel_width=30):
    f = np.ascontiguousarray(field, dtype=np.float64).copy()
    n = f.shape[-3:]
    _chk(lib().orc_smooth_field_f64(_p(f), _i3((n[2], n[1], n[0])), _d3(sigma_vox), C.c_double(max_error),
                                    C.c_int(max_kernel_width)), "smooth_field")
    return f


def warp_image(moving, field, out_like=None, edge_value=np.finfo(np.float32).max):
    m = np.ascontiguousarray(moving.arr, dtype=np.float32)
    ref = out_like or moving
    f = np.ascontiguousarray(field, dtype=np.float64)
    out = np.empty(ref.arr.shape[-3:], dtype=np.float32)
    gm, go = moving.geom(), ref.geom()
    _chk(lib().orc_warp_image_f32(_p(m), C.byref(gm), _p(f), C.byref(go), C.c_float(edge_value), _p(out)), "warp")
    return ref.like(out)


def esm_update(fixed, warped, max_step_length=0.5, intensity_threshold=0.001, denominator_threshold=1e-9):
    f = np.ascontiguousarray(fixed.arr, dtype=np.float32)
    w = np.ascontiguousarray(warped.arr, dtype=np.float32)
    upd = np.empty((3,) + f.shape, dtype=np.float64)
    st = DemonsStats()
    g = fixed.geom()
    _chk(lib().orc_esm_update(_p(f), _p(w), C.byref(g), C.c_double(max_step_length), C.c_double(intensity_threshold),
                              C.c_double(denominator_threshold), _p(upd), C.byref(st)), "esm_update")
    return upd, st


class DemonsFilter:
    """sitk.FastSymmetricForcesDemonsRegistrationFilter with SimpleITK 2.3.1's defaults, exposing the
    duck-typed protocol multiscale_demons needs (platipy/imaging/registration/deformable.py:144,149,157)."""

    def __init__(self):
        self.standard_deviations = [1.0, 1.0, 1.0]
        self.update_field_standard_deviations = [1.0, 1.0, 1.0]
        self.number_of_iterations = 10
        self.maximum_rms_error = 0.02
        self.maximum_update_step_length = 0.5
        self.smooth_displacement_field = True
        self.smooth_update_field = False
        self.maximum_kernel_width = 30
        self.maximum_error = 0.1
        self.intensity_difference_threshold = 0.001
        self.stats = None

    def SetNumberOfIterations(self, n):
        self.number_of_iterations = int(n)

    def SetStandardDeviations(self, s):
        self.standard_deviations = [float(x) for x in np.broadcast_to(np.asarray(s, dtype=float), (3,))]

    def GetStandardDeviations(self):
        return tuple(self.standard_deviations)

    def SetSmoothUpdateField(self, b):
        self.smooth_update_field = bool(b)

    def SetSmoothDisplacementField(self, b):
        self.smooth_displacement_field = bool(b)

    def SetMaximumRMSError(self, v):
        self.maximum_rms_error = float(v)

    def GetElapsedIterations(self):
        return self.stats.elapsed_iterations

    def GetMetric(self):
        return self.stats.metric

    def GetRMSChange(self):
        return self.stats.rms_change

    def Execute(self, fixed, moving):
        f = np.ascontiguousarray(fixed.arr, dtype=np.float32)
        m = np.ascontiguousarray(moving.arr, dtype=np.float32)
        field = np.empty((3,) + f.shape, dtype=np.float64)
        st = DemonsStats()
        g = fixed.geom()
        _chk(
            lib().orc_demons_execute(
                _p(f), _p(m), C.byref(g), C.c_int(self.number_of_iterations), _d3(self.standard_deviations),
                _d3(self.update_field_standard_deviations), C.c_int(int(self.smooth_displacement_field)),
                C.c_int(int(self.smooth_update_field)), C.c_double(self.maximum_rms_error),
                C.c_double(self.maximum_update_step_length), C.c_double(self.intensity_difference_threshold),
                C.c_double(self.maximum_error), C.c_int(self.maximum_kernel_width), _p(field), C.byref(st)),
            "demons_execute",
        )
        self.stats = st
        return fixed.like(field)


def _resample_common(fn, arr, dtype, vin, ref, affine, field_vol, interp, default_value):
    a = np.ascontiguousarray(arr, dtype=dtype)
    out = np.empty(ref.arr.shape[-3:], dtype=dtype)
    gi, go = vin.geom(), ref.geom()
    A = t = None
    if affine is not None:
        A = np.ascontiguousarray(affine[0], dtype=np.float64).reshape(9)
        t = np.ascontiguousarray(affine[1], dtype=np.float64).reshape(3)
    f = gd = None
    if field_vol is not None:
        f = np.ascontiguousarray(field_vol.arr, dtype=np.float64)
        gd = field_vol.geom()
    _chk(fn(_p(a), C.byref(gi), C.byref(go), _p(A) if A is not None else None, _p(t) if t is not None else None,
            _p(f) if f is not None else None, C.byref(gd) if gd is not None else None, C.c_int(interp),
            C.c_double(default_value), _p(out)), "resample")
    return ref.like(out)


def resample(vol, ref, affine=None, field_vol=None, interp=INTERP_LINEAR, default_value=0.0):
    """sitk.Resample / ResampleImageFilter for scalar images.  affine=(A 3x3, t) maps output physical
    points to input physical points; field_vol is a DisplacementFieldTransform's field (fp64, planar)."""
    if vol.arr.dtype == np.uint8:
        return _resample_common(lib().orc_resample_u8, vol.arr, np.uint8, vol, ref, affine, field_vol, interp,
                                default_value)
    return _resample_common(lib().orc_resample_f32, vol.arr, np.float32, vol, ref, affine, field_vol, interp,
                            default_value)


def resample_vec(field_vol, ref, through=None):
    """sitk.Resample on a VectorFloat64 image (linear, default 0); `through` is an optional displacement
    field transform's field (deformable.py:154)."""
    a = np.ascontiguousarray(field_vol.arr, dtype=np.float64)
    out = np.empty((3,) + tuple(ref.arr.shape[-3:]), dtype=np.float64)
    gi, go = field_vol.geom(), ref.geom()
    f = gd = None
    if through is not None:
        f = np.ascontiguousarray(through.arr, dtype=np.float64)
        gd = through.geom()
    _chk(lib().orc_resample_vec_f64(_p(a), C.byref(gi), C.byref(go), _p(f) if f is not None else None,
                                    C.byref(gd) if gd is not None else None, _p(out)), "resample_vec")
    return ref.like(out)


def recursive_gaussian_vec(field_vol, sigma):
    f = np.ascontiguousarray(field_vol.arr, dtype=np.float64).copy()
    g = field_vol.geom()
    _chk(lib().orc_recursive_gaussian_vec_f64(_p(f), C.byref(g), _d3(sigma)), "recursive_gaussian_vec")
    return field_vol.like(f)


def recursive_gaussian(vol, sigma):
    a = np.ascontiguousarray(vol.arr, dtype=np.float32)
    out = np.empty_like(a)
    g = vol.geom()
    _chk(lib().orc_recursive_gaussian_f32(_p(a), _p(out), C.byref(g), _d3(sigma)), "recursive_gaussian")
    return vol.like(out)


def recursive_gaussian_pass(vol, axis, sigma, order=0, normalize_across_scale=False):
    """One directional itk::RecursiveGaussianImageFilter (order 0: Gaussian, 1: first derivative per voxel) -> float32 Vol."""
    a = np.ascontiguousarray(vol.arr, dtype=np.float32)
    out = np.empty_like(a)
    g = vol.geom()
    _chk(lib().orc_recursive_gaussian_pass_f32(_p(a), _p(out), C.byref(g), C.c_int(int(axis)), C.c_double(float(sigma)),
                                               C.c_int(int(order)), C.c_int(int(bool(normalize_across_scale)))), "recursive_gaussian_pass")
    return vol.like(out)


def gradient_recursive_gaussian(vol, sigma=None, normalize_across_scale=True, use_image_direction=True):
    """itk::GradientRecursiveGaussianImageFilter as itk::ImageToImageMetricv4 configures its default moving-image gradient filter
    (reached from registration.Execute, platipy/imaging/registration/linear.py:238): sigma = the image's largest spacing,
    NormalizeAcrossScale on, UseImageDirection on.  Per output component d: the first-order filter along d, then the zero-order
    filters along the remaining axes in increasing order (float images between the filters), the result divided by spacing[d];
    the vector is then rotated by the direction cosines.  -> [3, Z, Y, X] float32, intensity per mm (times sigma).
    (ITK 5.3 from memory: parity unpinned.)"""
    sp = np.asarray(vol.spacing, dtype=np.float64)
    if sigma is None:
        sigma = float(sp.max())
    comps = []
    for d in range(3):
        cur = recursive_gaussian_pass(vol, d, sigma, 1, normalize_across_scale)
        for ax in range(3):
            if ax != d:
                cur = recursive_gaussian_pass(cur, ax, sigma, 0, normalize_across_scale)
        comps.append((cur.arr.astype(np.float64) / sp[d]).astype(np.float32))
    g = np.stack(comps)
    if use_image_direction:
        D = np.asarray(vol.direction, dtype=np.float64).reshape(3, 3)
        if not np.array_equal(D, np.eye(3)):
            g = np.einsum("rc,czyx->rzyx", D, g.astype(np.float64)).astype(np.float32)
    return g


# --------------------------------------------------------------------------------------
# platipy/imaging/registration/utils.py:195-267


def smooth_and_resample(vol, isotropic_voxel_size_mm=None, shrink_factor=None, smoothing_sigma=None,
                        interpolator=INTERP_LINEAR):
    image = vol
    if smoothing_sigma:
        if hasattr(smoothing_sigma, "__iter__"):
            smoothing_variance = [i * i for i in smoothing_sigma]
        else:
            smoothing_variance = (smoothing_sigma ** 2,) * 3
        maximum_kernel_width = int(max([8 * j * i for i, j in zip(image.spacing, smoothing_variance)]))
        image = discrete_gaussian(image, smoothing_variance, maximum_kernel_width)
    original_spacing = image.spacing
    original_size = image.size
    if shrink_factor and isotropic_voxel_size_mm:
        raise AttributeError("Function must be called with either isotropic_voxel_size_mm or shrink_factor, not both.")
    elif isotropic_voxel_size_mm:
        scale_factor = isotropic_voxel_size_mm * np.ones(3) / np.array(image.spacing)
        new_size = [int(sz / float(sf) + 0.5) for sz, sf in zip(original_size, scale_factor)]
    elif shrink_factor:
        if isinstance(shrink_factor, list):
            new_size = [int(sz / float(sf) + 0.5) for sz, sf in zip(original_size, shrink_factor)]
        else:
            new_size = [int(sz / float(shrink_factor) + 0.5) for sz in original_size]
    else:
        return image
    new_spacing = [((so - 1) * sp) / (sn - 1) for so, sp, sn in zip(original_size, original_spacing, new_size)]
    ref = Vol(np.zeros((new_size[2], new_size[1], new_size[0]), dtype=np.float32), new_spacing, image.origin,
              image.direction)
    return resample(image, ref, interp=interpolator, default_value=0.0)


# platipy/imaging/registration/deformable.py:31-187


def multiscale_demons(registration_algorithm, fixed_image, moving_image, initial_displacement_field=None,
                      isotropic_resample=None, resolution_staging=None, smoothing_sigmas=None, iteration_staging=None,
                      interp_order=INTERP_LINEAR, trace=None):
    fixed_images, moving_images = [], []
    for resolution, smoothing_sigma in zip(resolution_staging, smoothing_sigmas):
        iso = resolution if isotropic_resample else None
        shr = None if isotropic_resample else resolution
        fixed_images.append(smooth_and_resample(fixed_image, iso, shr, smoothing_sigma, interp_order))
        moving_images.append(smooth_and_resample(moving_image, iso, shr, smoothing_sigma, interp_order))

    if initial_displacement_field is None:
        nz, ny, nx = fixed_image.arr.shape
        initial = fixed_image.like(np.zeros((3, nz, ny, nx), dtype=np.float64))
    else:
        initial = resample_vec(initial_displacement_field, fixed_image)
    dvf_total = resample_vec(initial, fixed_image)  # :130

    for i in range(len(fixed_images)):
        f_image, m_image = fixed_images[i], moving_images[i]
        dvf_total = resample_vec(dvf_total, f_image)  # :137
        # :139-140 -- sitk.Resample(m_image, tfm_total, interp_order): default pixel value 0
        m_image = resample(m_image, m_image, field_vol=dvf_total, interp=interp_order, default_value=0.0)
        registration_algorithm.SetNumberOfIterations(iteration_staging[i])
        dvf_iter = registration_algorithm.Execute(f_image, m_image)  # :149
        # :154 -- dvf_total + Resample(dvf_iter, tfm_total)
        comp = resample_vec(dvf_iter, dvf_iter, through=dvf_total)
        dvf_total = dvf_total.like(dvf_total.arr + comp.arr)
        sigma = registration_algorithm.GetStandardDeviations()  # :157
        dvf_total = recursive_gaussian_vec(dvf_total, sigma)  # :158
        if trace is not None:
            trace.append({"level": i, "dvf_iter": dvf_iter, "dvf_total": dvf_total, "moving": m_image,
                          "fixed": f_image, "elapsed": registration_algorithm.GetElapsedIterations(),
                          "metric": registration_algorithm.GetMetric()})
    return resample_vec(dvf_total, fixed_image)  # :185


# platipy/imaging/registration/deformable.py:190-306


def fast_symmetric_forces_demons_registration(fixed_image, moving_image, resolution_staging=(8, 4, 1),
                                              iteration_staging=(10, 10, 10), isotropic_resample=False,
                                              initial_displacement_field=None, regularisation_kernel_mm=1.5,
                                              smoothing_sigma_factor=1, smoothing_sigmas=False, default_value=None,
                                              interp_order=INTERP_LINEAR, trace=None):
    moving_dtype = moving_image.arr.dtype
    fixed_image = fixed_image.like(fixed_image.arr.astype(np.float32))
    moving_image = moving_image.like(moving_image.arr.astype(np.float32))
    reg = DemonsFilter()
    reg.SetSmoothUpdateField(True)
    reg.SetSmoothDisplacementField(True)
    reg.SetStandardDeviations((np.array(regularisation_kernel_mm) / np.array(fixed_image.spacing)).tolist())
    if not smoothing_sigmas:
        smoothing_sigmas = [i * smoothing_sigma_factor for i in resolution_staging]
    dvf = multiscale_demons(reg, fixed_image, moving_image, resolution_staging=list(resolution_staging),
                            smoothing_sigmas=smoothing_sigmas, iteration_staging=list(iteration_staging),
                            isotropic_resample=isotropic_resample, initial_displacement_field=initial_displacement_field,
                            interp_order=interp_order, trace=trace)
    if default_value is None:
        default_value = 0
        if moving_image.arr.min() <= -1000:
            default_value = -1000
    registered = resample(moving_image, fixed_image, field_vol=dvf, interp=interp_order, default_value=default_value)
    registered = fixed_image.like(_cast(registered.arr, moving_dtype))
    return registered, dvf, dvf


def _cast(arr, dtype):
    """sitk.Cast from float to an integer pixel type truncates toward zero (static_cast)."""
    if np.issubdtype(dtype, np.integer):
        return np.trunc(arr).astype(dtype)
    return arr.astype(dtype)


# platipy/imaging/registration/utils.py:148-192


def apply_transform(input_image, reference_image=None, affine=None, field_vol=None, default_value=0,
                    interpolator=INTERP_NEAREST):
    ref = reference_image if reference_image is not None else input_image
    dt = input_image.arr.dtype
    src = input_image if dt == np.uint8 else input_image.like(input_image.arr.astype(np.float32))
    out = resample(src, ref, affine=affine, field_vol=field_vol, interp=interpolator, default_value=default_value)
    return ref.like(_cast(out.arr, dt))


# --------------------------------------------------------------------------------------
# platipy/imaging/label/fusion.py


def compute_weight_map(target, moving, vote_type="unweighted", vote_params=None):
    """fusion.py:56-202 (unweighted / global / local / block)."""
    p = {"sigma": 2.0, "epsilon": 1e-5, "factor": 1e12, "gain": 6, "blockSize": 5, "normalise": False}
    if vote_params:
        p.update(vote_params)
    t = target.arr.astype(np.float32)
    m = moving.arr.astype(np.float32)
    vt = vote_type.lower()
    if vt == "unweighted":
        return target.like((t * np.float32(0.0) + np.float32(1.0)).astype(np.float32))
    d = t.astype(np.float64) - m.astype(np.float64)
    sq = (d * d).astype(np.float32)
    if vt == "global":
        gw = p["factor"] / sq.sum(dtype=np.float64)  # :154-161 (np.float -> fp64 sum, quirk N5)
        return target.like((t * np.float32(0.0) + np.float32(gw)).astype(np.float32))
    if vt == "local":
        w = np.empty_like(t)
        _chk(lib().orc_weight_map_local(_p(t), _p(m), _i3(target.size), _d3(target.spacing), C.c_double(p["sigma"]),
                                        C.c_double(p["epsilon"]), _p(w)), "weight_map_local")
        if isinstance(p["normalise"], bool) and p["normalise"]:
            w = w / w.max()
        return target.like(w.astype(np.float32))
    if vt == "block":
        from scipy.ndimage import uniform_filter  # sitk.BoxMean(radius) = mean over (2r+1)^3, ZeroFluxNeumann

        bs = p["blockSize"]
        bs = (bs,) * 3 if isinstance(bs, int) else tuple(bs)
        raw = uniform_filter(sq.astype(np.float64), size=[2 * b + 1 for b in bs[::-1]], mode="nearest").astype(np.float32)
        w = (p["factor"] * (raw.astype(np.float64) ** -1.0).astype(np.float32).astype(np.float64) ** abs(p["gain"] / 2.0))
        w = w.astype(np.float32)
        if isinstance(p["normalise"], bool) and p["normalise"]:
            w = w / w.max()
        return target.like(w)
    raise ValueError(vote_type)


def combine_labels(atlas_set, structure_name, label="DIR", threshold=1e-4, smooth_sigma=1.0):
    """fusion.py:239-292.  atlas_set[case][label][name] are Vol objects."""
    case_id_list = list(atlas_set.keys())
    names = [structure_name] if isinstance(structure_name, str) else list(structure_name)
    out = {}
    for s_name in names:
        valid = [i for i in case_id_list if s_name in atlas_set[i][label].keys()]
        weights = [atlas_set[c][label]["Weight Map"].arr.astype(np.float32) for c in valid]
        wsum = weights[0]
        for w in weights[1:]:
            wsum = wsum + w  # float32 left fold, as functools.reduce over sitk images (:263)
        wsum = np.where(wsum == 0, np.float32(1), wsum)  # :264-266
        wl = [w * atlas_set[c][label][s_name].arr.astype(np.float32) for w, c in zip(weights, valid)]
        acc = wl[0]
        for x in wl[1:]:
            acc = acc + x
        comb = (acc / wsum).astype(np.float32)
        ref = atlas_set[valid[0]][label]["Weight Map"]
        comb = discrete_gaussian(ref.like(comb), smooth_sigma * smooth_sigma).arr  # :279
        comb = rescale_intensity(comb, 0.0, 1.0)  # :282
        if threshold:
            # sitk.Threshold(lower, upper=1, outsideValue=0): values outside [lower, upper] -> 0
            comb = np.where((comb < np.float32(threshold)) | (comb > np.float32(1.0)), np.float32(0), comb)
        out[s_name] = ref.like(comb.astype(np.float32))
    return out


def rescale_intensity(a, out_min, out_max):
    """itkRescaleIntensityImageFilter on a float image: scale/shift in RealType (double)."""
    a = a.astype(np.float32)
    imin, imax = float(a.min()), float(a.max())
    if imin != imax:
        scale = (float(out_max) - float(out_min)) / (imax - imin)
    elif imax != 0.0:
        scale = (float(out_max) - float(out_min)) / imax
    else:
        scale = 0.0
    shift = float(out_min) - imin * scale
    r = a.astype(np.float64) * scale + shift
    r = np.clip(r, out_min, out_max)  # IntensityLinearTransform clamps to [min, max]
    return r.astype(np.float32)


def binary_fillhole(mask_vol):
    """sitk.BinaryFillhole(mask) with defaults (fusion.py:308): background reachable from the image border through faces stays
    background (fullyConnected=False), every other background voxel becomes foreground."""
    from scipy import ndimage

    return mask_vol.like(ndimage.binary_fill_holes(mask_vol.arr != 0).astype(np.uint8))


def connected_component(mask_vol):
    """sitk.ConnectedComponent(mask) with defaults (fusion.py:311): face connectivity; [ITK-upstream] labels 1..n in raster order
    of each component's first voxel.  -> int32 label volume."""
    from scipy import ndimage

    lab, _ = ndimage.label(mask_vol.arr != 0)
    return mask_vol.like(lab.astype(np.int32))


def process_probability_image(prob, threshold=0.5):
    """fusion.py:295-328: /max -> BinaryThreshold(>= thr) -> BinaryFillhole -> ConnectedComponent -> largest."""
    from scipy import ndimage

    a = prob.arr.astype(np.float32)
    a = (a.astype(np.float64) / float(a.max())).astype(np.float32)   # Div functor: fp64 quotient, fp32 pixel
    b = a.astype(np.float64) >= float(threshold)                       # BinaryThreshold: lower <= pixel
    b = ndimage.binary_fill_holes(b)  # face connectivity background, as BinaryFillhole (fullyConnected=False)
    lab, n = ndimage.label(b)  # face connectivity (fullyConnected=False)
    if n == 0:
        return prob.like(b.astype(np.uint8))
    counts = np.bincount(lab.ravel())[1:]
    # ITK labels components in raster order of their first pixel; np.argmax picks the first maximal one
    best = 1 + int(np.argmax(counts))
    return prob.like((lab == best).astype(np.uint8))


# --------------------------------------------------------------------------------------
# binary morphology with ITK's ball (registration/utils.py:328-329; multiatlas/run.py:421-423)


def ball_element(radius):
    """[ITK-upstream FlatStructuringElement::Ball(radius), radiusIsParametric = False] as a [Z][Y][X] bool array:
    voxel offset d is in the element when sum_i (d_i / (r_i + 0.5))^2 <= 1 (ellipsoid of axes 2 r_i + 1 around
    the centre voxel, pixel-centre inclusion)."""
    rx, ry, rz = (int(r) for r in radius)
    z, y, x = np.mgrid[-rz:rz + 1, -ry:ry + 1, -rx:rx + 1].astype(np.float64)
    s = (x / (rx + 0.5)) ** 2
    s = s + (y / (ry + 0.5)) ** 2
    s = s + (z / (rz + 0.5)) ** 2
    return s <= 1.0


def binary_dilate_ball(mask_vol, radius):
    """sitk.BinaryDilate(mask, radius) with defaults: ball kernel, background outside the image."""
    from scipy import ndimage

    out = ndimage.binary_dilation(mask_vol.arr != 0, structure=ball_element(radius), border_value=0)
    return mask_vol.like(out.astype(np.uint8))


def binary_erode_ball(mask_vol, radius):
    """sitk.BinaryErode(mask, radius) with defaults: ball kernel, boundaryToForeground = True."""
    from scipy import ndimage

    out = ndimage.binary_erosion(mask_vol.arr != 0, structure=ball_element(radius), border_value=1)
    return mask_vol.like(out.astype(np.uint8))


def binary_closing_ball(mask_vol, radius):
    """sitk.BinaryMorphologicalClosing(mask, radius) with defaults (safeBorder = True): pad by the radius with
    background, dilate, erode, crop."""
    rx, ry, rz = (int(r) for r in radius)
    padded = Vol(np.pad(mask_vol.arr != 0, [(rz, rz), (ry, ry), (rx, rx)]).astype(np.uint8), mask_vol.spacing, mask_vol.origin)
    a = binary_erode_ball(binary_dilate_ball(padded, radius), radius).arr
    a = a[rz:a.shape[0] - rz, ry:a.shape[1] - ry, rx:a.shape[2] - rx]
    return mask_vol.like(np.ascontiguousarray(a))


# --------------------------------------------------------------------------------------
# platipy/imaging/label/projection.py:67-92 helpers


def maurer_distance_map(mask_vol, signed=True, inside_positive=False):
    """sitk.SignedMaurerDistanceMap(mask, squaredDistance=False, useImageSpacing=True): exact Euclidean distance
    (mm) to the nearest BORDER voxel of the object -- object voxels with a background voxel in their
    26-neighbourhood (itkSignedMaurerDistanceMapImageFilter extracts it with BinaryContourImageFilter,
    FullyConnected on) -- zero on the border, negative inside unless inside_positive."""
    from scipy import ndimage

    obj = mask_vol.arr != 0
    bg_near = ndimage.binary_dilation(~obj, structure=np.ones((3, 3, 3), bool))   # voxels outside the image are ignored
    border = obj & bg_near
    d = ndimage.distance_transform_edt(~border, sampling=mask_vol.spacing[::-1])
    if signed:
        d = np.where(obj != bool(inside_positive), -d, d)
    return mask_vol.like(d.astype(np.float32))


def label_contour(mask_vol):
    """sitk.LabelContour(mask) (fullyConnected=False): object voxels with a face neighbour of another value."""
    obj = mask_vol.arr != 0
    p = np.pad(obj, 1, mode="edge")   # a voxel outside the image never differs
    diff = np.zeros_like(obj)
    for ax in range(3):
        for sh in (0, 2):
            sl = [slice(1, -1)] * 3
            sl[ax] = slice(sh, p.shape[ax] - 2 + sh)
            diff |= p[tuple(sl)] != obj
    return mask_vol.like((obj & diff).astype(np.uint8))


def evaluate_distance_to_reference(reference_volume, test_volume, resample_factor=1):
    """projection.py:67-92"""
    d = np.abs(maurer_distance_map(test_volume, signed=True).arr)
    pts = label_contour(reference_volume).arr == 1
    return d[pts][::resample_factor]


def iar_q_values(atlas_set, reference_structure, label="DIR"):
    """The per-atlas Q metric of one run_iar pass (iar.py:91-229, MAD statistic, no spherical projection)."""
    from scipy.optimize import curve_fit
    from scipy.stats import norm

    ids = list(atlas_set.keys())
    prob = combine_labels(atlas_set, reference_structure, label=label)[reference_structure]
    rf = 5 if len(ids) < 12 else 1
    ref = process_probability_image(prob, 0.95)
    g = [evaluate_distance_to_reference(ref, process_probability_image(atlas_set[i][label][reference_structure], 0.1), rf) for i in ids]
    q = {}
    for k, cid in enumerate(ids):
        rest = g[:k] + g[k + 1:]
        med = np.median(rest, axis=0)
        mad = 1.4826 * np.median(np.abs(rest - np.median(rest, axis=0)), axis=0)
        if np.any(mad == 0):
            mad[mad == 0] = np.median(mad)
        z = np.ravel((g[k] - med) / mad)
        dens, edges = np.histogram(z, bins=np.linspace(-15, 15, 501), density=True)
        c = (edges[1:] + edges[:-1]) / 2.0
        f = lambda x, a, m, s: a * norm.pdf(x, loc=m, scale=s)  # noqa: E731
        try:
            popt, _ = curve_fit(f=f, xdata=c, ydata=dens)
            diff = np.abs(dens - f(c, *popt))
        except (RuntimeError, ValueError):
            diff = np.abs(dens - f(c, 1, dens.mean(), dens.std()))
        trap = np.trapezoid if hasattr(np, "trapezoid") else np.trapz
        q[cid] = float(trap(diff * np.abs(c) ** 2, c))
    return q
