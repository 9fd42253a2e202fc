"""Volumes whose displacement field spans MORE THAN 2^32 BYTES: 512 x 512 x 1400 voxels -- a whole-body
CT -- is 367 Mvoxel, 1.47 GB per scalar image and 4.4 GB per three-component field.  Component 2 of a planar field crosses the
2^32-byte mark at z = 1288, so every kernel that forms a byte offset in 32 bits goes wrong from there on.  Until round 6 such
volumes fell back to the first kernel generation and no GPU test had ever run one.

  * `Execute`: the generation-2 fused kernels now take them (their BIG instances: 64-bit bases for the field arrays,
    pp_demons_fused2.h) -- bit-identical to generation 1 on the same volume, and equal to the ORACLE's Execute, within the
    tolerances of tests/test_kernels.py::test_demons_execute, on z-slabs that cover the first planes, the planes around the
    2^32-byte crossing and the last planes (two iterations reach 11 planes, the slabs carry a 16-plane margin);
  * warp / label propagation through a field, composition, the recursive Gaussian of a field and the fusion arithmetic, against
    the oracle (or torch, for the element-wise kernels) on the last slab.

The oracle never sees the whole volume (it would need ~40 GB of host memory): every comparison is slab-local, which is exact for
these stencils beyond the margin."""
import time

import numpy as np
import pytest
import torch

from platipy_amd import _lib
from tests.helpers import record_stats

pytestmark = pytest.mark.gpu

NX, NY, NZ = 512, 512, 1400
SHAPE, SPACING = (NZ, NY, NX), (1.0, 1.0, 1.0)
PLANE = NX * NY
MARGIN = 16
CROSS = (1 << 32) // 4 - 2 * NX * NY * NZ      # first voxel of component 2 whose byte offset needs 33 bits
Z_CROSS = CROSS // PLANE


@pytest.fixture(scope="module")
def ctx():
    return _lib.Context(0, torch.cuda.current_stream().cuda_stream)


@pytest.fixture(scope="module")
def pair(ctx):
    """A CT-like pair without the bench's 12 ellipsoid masks at this size: body + smooth texture + noise, moving = fixed seen
    through a smooth field of <= 4 mm."""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(99)
    x = torch.arange(NX, device=dev, dtype=torch.float32).view(1, 1, NX)
    y = torch.arange(NY, device=dev, dtype=torch.float32).view(1, NY, 1)
    z = torch.arange(NZ, device=dev, dtype=torch.float32).view(NZ, 1, 1)
    body = (((x - NX / 2) / (0.42 * NX)) ** 2 + ((y - NY / 2) / (0.40 * NY)) ** 2) < 1
    tex = torch.nn.functional.interpolate(torch.randn((1, 1, 44, 16, 16), device=dev, generator=g), size=SHAPE, mode="trilinear",
                                          align_corners=True)[0, 0]
    clean = torch.where(body, 200.0 * tex, torch.full((), -1000.0, device=dev)).contiguous()
    del tex, body
    fixed = clean + 5.0 * torch.randn(SHAPE, device=dev, generator=g)
    dvf = torch.nn.functional.interpolate(torch.randn((1, 3, 22, 8, 8), device=dev, generator=g), size=SHAPE, mode="trilinear",
                                          align_corners=True)[0].contiguous()
    dvf *= 4.0 / float(torch.sqrt((dvf ** 2).sum(0)).max())
    geom = _lib.make_geom((NX, NY, NZ), SPACING)
    moving = torch.empty_like(clean)
    ctx.warp(clean, dvf, geom, -1000.0, moving)
    ctx.sync()
    moving += 5.0 * torch.randn(SHAPE, device=dev, generator=g)
    del clean
    return fixed.contiguous(), moving.contiguous(), dvf, geom


def _slabs():
    """(name, first plane handed to the oracle, last + 1, first plane compared, last + 1 compared)"""
    return [("first", 0, 48 + MARGIN, 0, 48),
            ("crossing", Z_CROSS - 24 - MARGIN, Z_CROSS + 24 + MARGIN, Z_CROSS - 24, Z_CROSS + 24),
            ("last", NZ - 48 - MARGIN, NZ, NZ - 48, NZ)]


def test_the_volume_is_in_the_band():
    assert 3 * NX * NY * NZ * 4 > 1 << 32 and NX * NY * NZ * 4 < 1 << 32
    assert 0 < Z_CROSS < NZ - 24 - MARGIN, Z_CROSS


def test_execute_generation2_on_a_field_beyond_4_gib(ctx, pair, monkeypatch):
    from oracle import oracle as O

    fixed, moving, _, geom = pair
    N = NX * NY * NZ
    out, times = {}, {}
    for name, env in (("generation 2 (BIG instances)", {}), ("generation 1", {"PP_FUSED_GEN": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        p = ctx.default_demons_params()
        p.smooth_update, p.iterations, p.max_rms_error, p.variant = 1, 2, 0.0, _lib.DEMONS_FUSED
        p.sigma_d_vox[:] = [1.5, 1.5, 1.5]
        field = torch.zeros((3,) + SHAPE, device="cuda")
        ctx.demons_execute(fixed, moving, geom, p, field, want_stats=False)      # warm-up: workspace
        ctx.sync()
        p.iterations = 6
        t0 = time.perf_counter()
        ctx.demons_execute(fixed, moving, geom, p, field, want_stats=False)
        ctx.sync()
        times[name] = (time.perf_counter() - t0) / 6
        p.iterations = 2
        st = ctx.demons_execute(fixed, moving, geom, p, field)
        ctx.sync()
        out[name] = (field, st)
        for k in env:
            monkeypatch.delenv(k)
    (f2, st2), (f1, st1) = out["generation 2 (BIG instances)"], out["generation 1"]
    assert st2.elapsed_iterations == 2 == st1.elapsed_iterations and st2.n_pixels == st1.n_pixels
    same = bool(torch.equal(f2.view(torch.int32), f1.view(torch.int32)))
    del f1
    stats = {"size": [NX, NY, NZ], "field_bytes": 12 * N, "z_of_the_2^32_crossing_component_2": int(Z_CROSS),
             "ms_per_iteration": {k: 1e3 * v for k, v in times.items()}, "Mvoxels_per_s": {k: N / v / 1e6 for k, v in times.items()},
             "generation2_equals_generation1_bitwise": same, "slabs": {}}
    # the oracle, slab by slab
    for name, a, b, ca, cb in _slabs():
        fh, mh = fixed[a:b].cpu().numpy(), moving[a:b].cpu().numpy()
        flt = O.DemonsFilter()
        flt.SetSmoothUpdateField(True)
        flt.SetSmoothDisplacementField(True)
        flt.SetStandardDeviations([1.5, 1.5, 1.5])
        flt.SetNumberOfIterations(2)
        flt.SetMaximumRMSError(0.0)
        want = flt.Execute(O.Vol(fh, SPACING, (0.0, 0.0, float(a))), O.Vol(mh, SPACING, (0.0, 0.0, float(a)))).arr
        got = f2[:, ca:cb].cpu().numpy()
        err = np.abs(got - want[:, ca - a:cb - a])
        stats["slabs"][name] = {"planes": [ca, cb], "max_abs_mm": float(err.max()), "rms_mm": float(np.sqrt((err.astype(np.float64) ** 2).mean())),
                                "field_abs_max_mm": float(np.abs(got).max())}
    record_stats("huge_volume_execute_512x512x1400", stats)
    print("512 x 512 x 1400 Execute:", stats)
    assert same
    for name, s in stats["slabs"].items():
        assert s["max_abs_mm"] <= 2e-3 and s["rms_mm"] <= 5e-5 and s["field_abs_max_mm"] > 0.05, (name, s)


def test_once_per_level_kernels_on_the_last_slab_of_a_huge_volume(ctx, pair):
    from oracle import oracle as O

    fixed, moving, dvf, geom = pair
    N = NX * NY * NZ
    a, ca = NZ - 40 - 2 * MARGIN, NZ - 40
    org = (0.0, 0.0, float(a))
    fld = dvf[:, a:].cpu().numpy().astype(np.float64)
    fvol = O.Vol(fld, SPACING, org)
    stats = {}
    # warp through the field (linear) and label propagation (nearest, uint8): sitk.Resample(image, DisplacementFieldTransform)
    out = torch.empty_like(moving)
    ctx.resample(moving, geom, geom, out, field=dvf, interp=_lib.INTERP_LINEAR, default_value=-1000.0)
    want = O.resample(O.Vol(moving[a:].cpu().numpy(), SPACING, org), O.Vol(moving[a:].cpu().numpy(), SPACING, org), field_vol=fvol,
                      interp=O.INTERP_LINEAR, default_value=-1000.0).arr
    stats["warp_linear_max_abs"] = float(np.abs(out[ca:].cpu().numpy() - want[ca - a:]).max())
    del out
    mask = (fixed > -500).to(torch.uint8).contiguous()
    mout = torch.empty_like(mask)
    ctx.resample(mask, geom, geom, mout, field=dvf, interp=_lib.INTERP_NEAREST, default_value=0.0, u8=True)
    mwant = O.resample(O.Vol(mask[a:].cpu().numpy(), SPACING, org), O.Vol(mask[a:].cpu().numpy(), SPACING, org), field_vol=fvol,
                       interp=O.INTERP_NEAREST).arr
    stats["label_propagation_voxels_differing"] = int((mout[ca:].cpu().numpy() != mwant[ca - a:]).sum())
    del mask, mout
    # composition D += d o (id + D)   (deformable.py:154) and the level's recursive Gaussian (:158)
    small = (0.25 * dvf.flip(0)).contiguous()
    total = dvf.clone()
    ctx.compose_field(total, small, geom)
    sm = O.Vol(small[:, a:].cpu().numpy().astype(np.float64), SPACING, org)
    comp = O.resample_vec(sm, sm, through=fvol)
    stats["compose_max_abs_mm"] = float(np.abs(total[:, ca:].cpu().numpy() - (fld + comp.arr)[:, ca - a:]).max())
    del small
    ctx.recursive_gaussian_field(total, geom, [1.5, 1.5, 1.5])
    # (the oracle smooths the product's composed slab: its own lower boundary is 32 planes from the compared ones, 21 sigma)
    ctx.sync()
    ref_total = dvf.clone()
    ctx.compose_field(ref_total, (0.25 * dvf.flip(0)).contiguous(), geom)
    want = O.recursive_gaussian_vec(O.Vol(ref_total[:, a:].cpu().numpy().astype(np.float64), SPACING, org), [1.5, 1.5, 1.5]).arr
    stats["recursive_gaussian_max_abs_mm"] = float(np.abs(total[:, ca:].cpu().numpy() - want[:, ca - a:]).max())
    del ref_total, total
    # fusion arithmetic on the field-sized range (3 N floats: offsets beyond 2^32 bytes)
    w = torch.rand(3 * N, device="cuda") + 0.5
    lab = (torch.rand(3 * N, device="cuda") > 0.5).to(torch.uint8)
    wsum, wlsum = torch.zeros(3 * N, device="cuda"), torch.zeros(3 * N, device="cuda")
    ctx.fuse_accumulate(w, lab, wsum, wlsum, 3 * N)
    ctx.sync()
    tail = slice(3 * N - (1 << 22), 3 * N)
    stats["fuse_accumulate_tail_equal"] = bool(torch.equal(wsum[tail], w[tail]) and torch.equal(wlsum[tail], w[tail] * lab[tail].float()))
    res = torch.empty_like(w)
    ctx.fuse_divide(wlsum, wsum, res, 3 * N)
    ctx.sync()
    stats["fuse_divide_tail_max_abs"] = float((res[tail] - torch.where(wsum[tail] > 0, wlsum[tail] / wsum[tail], torch.zeros(()).cuda())).abs().max())
    record_stats("huge_volume_once_per_level_kernels", stats)
    print("512 x 512 x 1400, last slab:", stats)
    assert stats["warp_linear_max_abs"] <= 2e-3                 # HU-scale data, fp32 interpolation
    assert stats["label_propagation_voxels_differing"] == 0
    assert stats["compose_max_abs_mm"] <= 2e-5 and stats["recursive_gaussian_max_abs_mm"] <= 2e-5
    assert stats["fuse_accumulate_tail_equal"] and stats["fuse_divide_tail_max_abs"] <= 1e-6
