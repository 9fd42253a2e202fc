"""Drop-in for platipy/imaging/registration/deformable.py:31-306: multi-resolution fast
symmetric-forces demons.  Same function names, keyword arguments, defaults and return values; the
voxel-level work (pyramids, warps, the demons inner loop, field composition and regularisation) runs
in the HIP kernels behind include/platipy_amd.h, on fp32 volumes resident in HBM.

Deviations, all deliberate and visible:
  * the displacement field is computed in fp32 (the reference keeps sitkVectorFloat64); the tolerance against the
    fp64 restatement is stated and tested in tests/; `field_dtype=torch.float64` returns it in the reference's type;
  * `ncores` is accepted and ignored (it set ITK's CPU thread count);
  * iteration observers (`verbose=True`, AddCommand) fire after each level's Execute, once per iteration that ran, with the
    per-iteration metric the device recorded (the loop itself never returns to the host);
  * non-identity direction cosines (axis flips / oblique acquisitions): the registration runs in the image's
    own index-aligned frame -- every stage is linear in the field and the pyramid grids share one origin and
    one direction, so this is the same computation -- and the field is rotated back to physical (LPS)
    components at the end.  (ITK's behaviour for this case could not be checked: parity unpinned.)
"""
import contextlib
import weakref

import numpy as np
import torch

from .. import _lib
from ..image import Image, as_image, cast_tensor, to_sitk
from .. import runtime
from ..transform import DisplacementFieldTransform, sitkLinear
from .utils import resample_field, resample_image, smooth_and_resample, transform_to_displacement_field


class _IterationView:
    """The filter's measurements as an sitkIterationEvent observer sees them after iteration `elapsed_iterations`."""

    def __init__(self, elapsed, metric, rms_change):
        self.elapsed_iterations, self.metric, self.rms_change = elapsed, metric, rms_change


class HipDemonsFilter:
    """sitk.FastSymmetricForcesDemonsRegistrationFilter with SimpleITK's defaults, satisfying the
    duck-typed protocol multiscale_demons needs (reference deformable.py:144,149,157):
    SetNumberOfIterations / Execute / GetStandardDeviations."""

    def __init__(self, variant="auto"):
        self._standard_deviations = [1.0, 1.0, 1.0]
        self._update_field_standard_deviations = [1.0, 1.0, 1.0]
        self._iterations = 10
        self._max_rms = 0.02
        self._max_step = 0.5
        self._smooth_disp = True
        self._smooth_update = False
        self._max_kernel_width = 30
        self._max_error = 0.1
        self._intensity_threshold = 0.001
        self._variant = {"auto": _lib.DEMONS_AUTO, "staged": _lib.DEMONS_STAGED, "fused": _lib.DEMONS_FUSED}[variant]
        self._stats = None
        self._pending = None      # the context whose history ring holds the last Execute's measurements, until they are asked for
        self._commands = []

    # configuration, SimpleITK names
    def SetNumberOfThreads(self, n):  # CPU concept; kept for signature compatibility
        pass

    def SetNumberOfIterations(self, n):
        self._iterations = int(n)

    def GetNumberOfIterations(self):
        return self._iterations

    def SetStandardDeviations(self, s):
        self._standard_deviations = [float(v) for v in np.broadcast_to(np.asarray(s, dtype=np.float64), (3,))]

    def GetStandardDeviations(self):
        return tuple(self._standard_deviations)

    def SetUpdateFieldStandardDeviations(self, s):
        self._update_field_standard_deviations = [float(v) for v in np.broadcast_to(np.asarray(s, dtype=np.float64), (3,))]

    def SetSmoothUpdateField(self, b):
        self._smooth_update = bool(b)

    def SetSmoothDisplacementField(self, b):
        self._smooth_disp = bool(b)

    def SetMaximumRMSError(self, v):
        self._max_rms = float(v)

    def SetMaximumUpdateStepLength(self, v):
        self._max_step = float(v)

    def SetIntensityDifferenceThreshold(self, v):
        self._intensity_threshold = float(v)

    def SetMaximumKernelWidth(self, v):
        self._max_kernel_width = int(v)

    def SetMaximumError(self, v):
        self._max_error = float(v)

    def AddCommand(self, event, fn):
        """registration_method.AddCommand(sitk.sitkIterationEvent, fn) (reference deformable.py:260-264).  The loop runs on the
        device without host round trips, so the observers fire after Execute -- once per iteration that ran, in order, with
        GetElapsedIterations() / GetMetric() / GetRMSChange() answering what they answered at that iteration's event (the
        kernel that closes an iteration keeps the values in a device ring, pp_demons_history)."""
        self._commands.append(fn)

    # measurements.  Execute does not read them back: that is a host-device round trip per pyramid level with the GPU
    # idle behind it.  They stay in the context's history ring (pp_demons_history) until one of the getters asks, or until
    # ANOTHER filter is about to run on the same context (its Execute resolves this one's first).
    def _resolve(self):
        ctx, self._pending = self._pending, None
        owner_ref = getattr(ctx, "_demons_owner", None) if ctx is not None else None
        if owner_ref is not None and owner_ref() is self:
            ctx._demons_owner = None
            history = ctx.demons_history()
            self._stats = _IterationView(len(history), *history[-1]) if history else None

    def GetElapsedIterations(self):
        self._resolve()
        return self._stats.elapsed_iterations if self._stats else 0

    def GetMetric(self):
        self._resolve()
        return self._stats.metric if self._stats else float("nan")

    def GetRMSChange(self):
        self._resolve()
        return self._stats.rms_change if self._stats else float("nan")

    def Execute(self, fixed_image, moving_image):
        """registration_algorithm.Execute(f_image, m_image) (reference deformable.py:149).  Given platipy_amd Images it
        returns a planar fp32 vector Image in HBM; given SimpleITK images -- the reference's own multiscale_demons calling
        this filter -- it returns what sitk's filter returns there, a VectorFloat64 sitk.Image on the fixed grid, which
        the caller's sitk.Resample(dvf_iter, tfm_total) (:154) consumes."""
        wants_sitk = not isinstance(fixed_image, Image)
        f, m = as_image(fixed_image), as_image(moving_image)
        if f.GetSize() != m.GetSize():
            raise ValueError("demons: fixed and moving image must be on the same grid (reference deformable.py:210-211)")
        # Direction cosines other than the identity (axis flips, oblique acquisitions): sitk's filter takes them
        # (deformable.py:149 hands it whatever grid the pyramid has).  Both images are on ONE grid, so in that grid's
        # index-aligned frame the iteration is the identity-direction one on the same voxel arrays -- the image gradients,
        # hence the update, rotate with the frame (|J| and the normaliser do not change), and the component-wise smoothing
        # commutes with a constant rotation -- and the physical (LPS) field is R times the local one.
        oriented = f.direction != _IDENTITY
        if oriented and m.direction != f.direction:
            raise ValueError("demons: fixed and moving image must be on the same grid (reference deformable.py:210-211)")
        ctx = runtime.context(f.device)
        ft = (f.tensor if f.tensor.dtype == torch.float32 else f.tensor.float()).contiguous()
        mt = (m.tensor if m.tensor.dtype == torch.float32 else m.tensor.float()).contiguous()
        p = ctx.default_demons_params()
        p.iterations = self._iterations
        p.sigma_d_vox[:] = self._standard_deviations
        p.sigma_u_vox[:] = self._update_field_standard_deviations
        p.smooth_displacement = int(self._smooth_disp)
        p.smooth_update = int(self._smooth_update)
        p.max_rms_error = self._max_rms
        p.max_step_length = self._max_step
        p.intensity_threshold = self._intensity_threshold
        p.max_error = self._max_error
        p.max_kernel_width = self._max_kernel_width
        p.variant = self._variant
        field = torch.empty((3,) + f.shape, dtype=torch.float32, device=ft.device)
        # (the context remembers the filter whose measurements it still holds through a WEAK reference: once that filter is
        # gone nobody can ask for them, and reading them back -- a stream synchronisation in front of this Execute, 0.9 ms of
        # host stall at the start of every registration after the first -- would be for nothing)
        owner_ref = getattr(ctx, "_demons_owner", None)
        owner = owner_ref() if owner_ref is not None else None
        if owner is not None and owner is not self:
            owner._resolve()
        self._pending = None
        # (the ring holds PP_HIST_CAP = 4096 iterations; beyond that, and with observers, the measurements are read back here)
        lazy = not self._commands and 0 < self._iterations <= 4096
        geom = Image(ft, f.spacing, f.origin, _IDENTITY).geom() if oriented else f.geom()
        final = ctx.demons_execute(ft, mt, geom, p, field, want_stats=not lazy)
        if oriented:
            field = _rotate_field(field, f.direction)
        if self._commands:
            history = ctx.demons_history()
            for k, (metric, rms) in enumerate(history):
                self._stats = _IterationView(k + 1, metric, rms)
                for fn in self._commands:
                    fn()
            # iterations beyond the ring's capacity (demons_history has warned): their events still fire, in order, with the
            # iteration count right and the last measurements that exist
            for k in range(len(history), int(getattr(ctx, "last_history_iterations", len(history)))):
                self._stats = _IterationView(k + 1, final.metric, final.rms_change)
                for fn in self._commands:
                    fn()
        self._stats = final
        ctx._demons_owner = weakref.ref(self) if lazy else None
        self._pending = ctx if lazy else None
        out = Image(field, f.spacing, f.origin, f.direction, True)
        return to_sitk(out) if wants_sitk else out


_IDENTITY = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)


_CT_PROBE_VOXELS = 1 << 20


def _rotate_field(field, direction, transpose=False):
    """R (or R^T) applied to the three components of a planar field tensor [3, Z, Y, X]."""
    R = torch.tensor(direction, dtype=field.dtype, device=field.device).reshape(3, 3)
    return torch.einsum("rc,czyx->rzyx", R.t() if transpose else R, field).contiguous()


def _local_frame(fixed_image, moving_image, initial_displacement_field=None):
    """The images (and an initial field) in the fixed image's index-aligned frame, q_local = R^T (q - o_f) + o_f: every
    index map is unchanged, an image whose own origin differs from the fixed one's sits at R^T (o - o_f) + o_f there, and a
    physical displacement d becomes R^T d.  Every stage of the registration is linear in the field and all pyramid grids
    share one origin and one direction, so the run in this frame is the same computation; its field is rotated back by R."""
    true_direction = fixed_image.direction
    if moving_image.direction != true_direction:
        raise NotImplementedError("demons: fixed and moving image must share their direction cosines")
    Rm = np.asarray(true_direction, dtype=np.float64).reshape(3, 3)
    o_f = np.asarray(fixed_image.origin, dtype=np.float64)

    def local_origin(o):
        return tuple(Rm.T @ (np.asarray(o, dtype=np.float64) - o_f) + o_f)

    moving_l = Image(moving_image.tensor, moving_image.spacing, local_origin(moving_image.origin), _IDENTITY)
    fixed_l = Image(fixed_image.tensor, fixed_image.spacing, fixed_image.origin, _IDENTITY)
    field_l = None
    if initial_displacement_field is not None:
        f0 = as_image(initial_displacement_field)
        if f0.direction != true_direction:
            raise NotImplementedError("demons: the initial displacement field must share the images' direction cosines")
        field_l = Image(_rotate_field(f0.tensor.float(), true_direction, transpose=True), f0.spacing, local_origin(f0.origin),
                        _IDENTITY, True)
    return fixed_l, moving_l, field_l


def _zero_field(reference):
    return Image(torch.zeros((3,) + reference.shape, dtype=torch.float32, device=reference.device), reference.spacing,
                 reference.origin, reference.direction, True)


def multiscale_demons(registration_algorithm, fixed_image, moving_image, initial_transform=None,
                      initial_displacement_field=None, isotropic_resample=None, resolution_staging=None,
                      smoothing_sigmas=None, iteration_staging=None, interp_order=sitkLinear, _exclusive=None):
    """Run `registration_algorithm` coarse-to-fine (reference deformable.py:31-187).  Any object with
    SetNumberOfIterations / Execute(fixed, moving) -> vector Image / GetStandardDeviations works.
    `_exclusive`: a contextlib.ExitStack of the caller's that receives runtime.exclusive() when the first throughput-bound
    level begins (several atlas chains on one GPU: runtime.Turnstile), so that the caller's own full-resolution work after this
    function stays inside the same section; without one the section ends with this function."""
    if _exclusive is None:
        with contextlib.ExitStack() as stack:
            return multiscale_demons(registration_algorithm, fixed_image, moving_image, initial_transform, initial_displacement_field,
                                     isotropic_resample, resolution_staging, smoothing_sigmas, iteration_staging, interp_order, stack)
    fixed_image, moving_image = as_image(fixed_image), as_image(moving_image)
    if fixed_image.direction != _IDENTITY:
        # a direct call with oriented images (fast_symmetric_forces_demons_registration converts before it calls): the loop
        # in the index-aligned frame, the field rotated back to physical components
        if initial_transform is not None:
            raise NotImplementedError("multiscale_demons: initial_transform with non-identity direction cosines; pass its "
                                      "displacement field as initial_displacement_field")
        true_direction = fixed_image.direction
        f_l, m_l, d_l = _local_frame(fixed_image, moving_image, initial_displacement_field)
        out = multiscale_demons(registration_algorithm, f_l, m_l, None, d_l, isotropic_resample, resolution_staging,
                                smoothing_sigmas, iteration_staging, interp_order, _exclusive)
        return Image(_rotate_field(out.tensor, true_direction), out.spacing, out.origin, true_direction, True)
    ctx = runtime.context(fixed_image.device)
    fixed_images, moving_images = [], []
    for resolution, smoothing_sigma in zip(resolution_staging, smoothing_sigmas):
        isotropic_voxel_size_mm = resolution if isotropic_resample else None
        shrink_factor = None if isotropic_resample else resolution
        # (_share_input: the levels are only ever read below -- warped, differenced, never written in place --, so an unsmoothed
        # shrink-1 level may be the caller's own tensor; tests/test_registration.py holds the inputs to be untouched)
        fixed_images.append(smooth_and_resample(fixed_image, isotropic_voxel_size_mm=isotropic_voxel_size_mm,
                                                shrink_factor=shrink_factor, smoothing_sigma=smoothing_sigma,
                                                interpolator=interp_order, _share_input=True))
        moving_images.append(smooth_and_resample(moving_image, isotropic_voxel_size_mm=isotropic_voxel_size_mm,
                                                 shrink_factor=shrink_factor, smoothing_sigma=smoothing_sigma,
                                                 interpolator=interp_order, _share_input=True))

    if initial_displacement_field is None:
        if initial_transform is not None:
            # :101-108 -- sitk.TransformToDisplacementField(initial_transform, VectorFloat64, fixed grid)
            dvf_total = transform_to_displacement_field(initial_transform, fixed_image)
        else:
            dvf_total = None        # the zero field of :110-123; resampled onto any grid it is that grid's zero field
    else:
        dvf_total = resample_field(as_image(initial_displacement_field), fixed_image)

    in_section = False
    for i in range(len(fixed_images)):
        f_image, m_image = fixed_images[i], moving_images[i]
        if not in_section and f_image.tensor.numel() >= runtime.HEAVY_VOXELS:
            _exclusive.enter_context(runtime.exclusive(f_image.device))      # (a no-op outside the multi-atlas worker threads)
            in_section = True
        if dvf_total is None:
            dvf_total = _zero_field(f_image)       # (not 800 MB of zeros at full resolution resampled onto the coarsest grid)
        else:
            dvf_total = resample_field(dvf_total, f_image, copy=False)                           # :137 (dvf_total is this loop's own)
        # :139-140 -- sitk.Resample(m_image, tfm_total, interp_order): default pixel value 0 (quirk N4)
        m_image = resample_image(m_image, m_image, DisplacementFieldTransform(dvf_total), interp_order, 0.0)
        registration_algorithm.SetNumberOfIterations(iteration_staging[i])
        dvf_iter = registration_algorithm.Execute(f_image, m_image)                              # :149
        ctx.compose_field(dvf_total.tensor, dvf_iter.tensor.contiguous(), f_image.geom())        # :154
        sigma = registration_algorithm.GetStandardDeviations()                                   # :157
        ctx.recursive_gaussian_field(dvf_total.tensor, f_image.geom(), sigma)                    # :158 (quirk N2)
    if dvf_total is None:
        return _zero_field(fixed_image)
    return resample_field(dvf_total, fixed_image, copy=False)                                    # :185


def fast_symmetric_forces_demons_registration(
    fixed_image,
    moving_image,
    resolution_staging=[8, 4, 1],
    iteration_staging=[10, 10, 10],
    isotropic_resample=False,
    initial_displacement_field=None,
    regularisation_kernel_mm=1.5,
    smoothing_sigma_factor=1,
    smoothing_sigmas=False,
    default_value=None,
    ncores=1,
    interp_order=sitkLinear,
    verbose=False,
    variant="auto",
    field_dtype=torch.float32,
):
    """Deformable image propagation using Fast Symmetric-Forces Demons (reference deformable.py:190-306).

    Returns (registered_image, output_transform, deformation_field), the field as a planar vector Image on the
    fixed grid.  The field is computed in fp32; `field_dtype=torch.float64` returns it as the reference does
    (sitkVectorFloat64, :97-98, :159 -- twice the memory, same values).  `variant` ("auto" | "fused" | "staged") picks
    the kernel schedule."""
    fixed_image, moving_image = as_image(fixed_image), as_image(moving_image)
    moving_image_type = moving_image.tensor.dtype
    fixed_image = fixed_image.astype(torch.float32)    # :236-241 (quirk N1: everything computes in float32)
    moving_image = moving_image.astype(torch.float32)

    identity = _IDENTITY
    true_direction = fixed_image.direction
    if true_direction != identity:
        fixed_image, moving_image, initial_displacement_field = _local_frame(fixed_image, moving_image, initial_displacement_field)

    registration_method = HipDemonsFilter(variant=variant)
    registration_method.SetNumberOfThreads(ncores)
    registration_method.SetSmoothUpdateField(True)
    registration_method.SetSmoothDisplacementField(True)
    regularisation_kernel_vox = np.array(regularisation_kernel_mm) / np.array(fixed_image.GetSpacing())
    registration_method.SetStandardDeviations(regularisation_kernel_vox.tolist())
    if verbose:
        registration_method.AddCommand(None, lambda: print("{0:3} = {1:10.5f}".format(
            registration_method.GetElapsedIterations(), registration_method.GetMetric())))
    if not smoothing_sigmas:
        smoothing_sigmas = [i * smoothing_sigma_factor for i in resolution_staging]

    if default_value is None:
        # CT-like moving image (:286-291).  Asked BEFORE the registration is enqueued: the answer is a host read-back, and
        # behind the whole registration it would idle the GPU until the host has come back and launched the final resample.
        mt = moving_image.tensor
        if mt.dtype == torch.float32 and mt.is_contiguous():
            # only "is any voxel <= -1000" matters: a CT's first planes are air, so the first 2^20 voxels usually answer it
            # (a 4 MB read); the whole volume is read only when they do not
            c = runtime.context(mt.device)
            head = min(mt.numel(), _CT_PROBE_VOXELS)
            lowest = c.minmax(mt, head)[0]
            if lowest > -1000 and head < mt.numel():
                lowest = c.minmax(mt, mt.numel())[0]
        else:
            lowest = float(mt.min())
        default_value = -1000 if lowest <= -1000 else 0

    with contextlib.ExitStack() as section:      # (runtime.exclusive from the first throughput-bound level to the final warp)
        deformation_field = multiscale_demons(
            registration_algorithm=registration_method, fixed_image=fixed_image, moving_image=moving_image,
            resolution_staging=resolution_staging, smoothing_sigmas=smoothing_sigmas, iteration_staging=iteration_staging,
            isotropic_resample=isotropic_resample, initial_displacement_field=initial_displacement_field,
            interp_order=interp_order, _exclusive=section)

        output_transform = DisplacementFieldTransform(deformation_field)
        registered_image = resample_image(moving_image, fixed_image, output_transform, interp_order, default_value)
    registered_image = registered_image.like(cast_tensor(registered_image.tensor, moving_image_type))
    if true_direction != identity:
        phys = _rotate_field(deformation_field.tensor, true_direction)
        deformation_field = Image(phys, deformation_field.spacing, deformation_field.origin, true_direction, True)
        output_transform = DisplacementFieldTransform(deformation_field)
        registered_image = Image(registered_image.tensor, registered_image.spacing, registered_image.origin, true_direction)
    if field_dtype != torch.float32:
        deformation_field = deformation_field.like(deformation_field.tensor.to(field_dtype), True)
        output_transform = DisplacementFieldTransform(deformation_field)
    return registered_image, output_transform, deformation_field
