"""BASELINE.json's full size (512x512x256) on the GPU: size-independent properties of the kernels (identities,
linearity, exact shifts), agreement of the two kernel schedules, and the bit-exact mask propagation against the
oracle (which resamples a 67-Mvoxel uint8 volume in about a second on the GPU box's host cores)."""
import numpy as np
import pytest
import torch

from platipy_amd import _lib

pytestmark = pytest.mark.gpu

NX, NY, NZ = 512, 512, 256
SHAPE = (NZ, NY, NX)
SPACING = (0.9766, 0.9766, 2.5)


@pytest.fixture(scope="module")
def env():
    from bench import synth_pair

    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    fixed, moving, _ = synth_pair(ctx, SHAPE, (1.0, 1.0, 1.0), 77, dev)
    return ctx, fixed, moving, _lib.make_geom((NX, NY, NZ), SPACING, (-250.0, -250.0, 10.0))


def test_identities_and_exact_shifts(env):
    ctx, fixed, moving, g = env
    out = torch.empty_like(moving)
    zero = torch.zeros((3,) + SHAPE, device="cuda")
    ctx.warp(moving, zero, g, 3.0e38, out)
    torch.cuda.synchronize()
    assert torch.equal(out, moving)                                   # warp by zero: identity, bit for bit
    g1 = _lib.make_geom((NX, NY, NZ), (1.0, 1.0, 2.0))                 # spacings whose reciprocals are exact in fp32
    sh = zero.clone()
    sh[0] = 3.0
    sh[2] = -4.0
    ctx.warp(moving, sh, g1, -5.0, out)
    torch.cuda.synchronize()
    assert torch.equal(out[2:, :, :-3], moving[:-2, :, 3:])           # integer-voxel displacement: exact shift
    assert bool((out[:2] == -5.0).all()) and bool((out[:, :, -3:] == -5.0).all())
    sh[0] = 3 * SPACING[0]                                            # 3 voxels at 0.9766 mm: 1/spacing is inexact in fp32
    sh[2] = -2 * SPACING[2]
    ctx.warp(moving, sh, g, -5.0, out)
    torch.cuda.synchronize()
    assert float((out[2:, :, :-4] - moving[:-2, :, 3:-1]).abs().max()) <= 2e-3
    # compose(D, 0) = D
    d = torch.randn((3,) + SHAPE, device="cuda")
    keep = d.clone()
    ctx.compose_field(d, zero, g)
    torch.cuda.synchronize()
    assert torch.equal(d, keep)
    # smoothing a constant field is the identity (taps sum to 1 up to rounding); a linear ramp is unchanged inside
    c = torch.full((3,) + SHAPE, 1.25, device="cuda")
    ctx.smooth_field(c, (NX, NY, NZ), [1.5 / s for s in SPACING])
    torch.cuda.synchronize()
    assert float((c - 1.25).abs().max()) <= 5e-7
    ramp = torch.arange(NX, device="cuda", dtype=torch.float32).view(1, 1, 1, NX).expand(3, NZ, NY, NX).contiguous() * 0.01
    r0 = ramp.clone()
    ctx.smooth_field(ramp, (NX, NY, NZ), [1.5 / s for s in SPACING])
    torch.cuda.synchronize()
    assert float((ramp - r0)[..., 4:-4].abs().max()) <= 5e-6


def test_smoothing_is_linear(env):
    ctx, _, _, _ = env
    size, sig = (NX, NY, NZ), [1.0, 1.0, 1.0]
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((3,) + SHAPE, device="cuda", generator=gen)
    y = torch.randn((3,) + SHAPE, device="cuda", generator=gen)
    z = 0.5 * x - 2.0 * y
    for t in (x, y, z):
        ctx.smooth_field(t, size, sig)
    torch.cuda.synchronize()
    assert float((z - (0.5 * x - 2.0 * y)).abs().max()) <= 2e-6


def test_fused_and_staged_schedules_agree(env):
    """Both schedules restate the same iteration; at full size they agree to fp32 rounding."""
    ctx, fixed, moving, _ = env
    g = _lib.make_geom((NX, NY, NZ), (1.0, 1.0, 1.0))
    res = {}
    for name, variant in (("fused", _lib.DEMONS_FUSED), ("staged", _lib.DEMONS_STAGED)):
        p = ctx.default_demons_params()
        p.iterations, p.smooth_update, p.smooth_displacement, p.max_rms_error, p.variant = 3, 1, 1, 0.0, variant
        p.sigma_d_vox[:] = [1.5, 1.5, 1.5]
        field = torch.empty((3,) + SHAPE, device="cuda")
        st = ctx.demons_execute(fixed, moving, g, p, field)
        res[name] = (field, st)
    (ff, sf), (fs, ss) = res["fused"], res["staged"]
    assert sf.elapsed_iterations == ss.elapsed_iterations == 3
    assert sf.n_pixels == ss.n_pixels == NX * NY * NZ
    np.testing.assert_allclose(sf.metric, ss.metric, rtol=1e-5)
    np.testing.assert_allclose(sf.rms_change, ss.rms_change, rtol=1e-5)
    diff = (ff - fs).abs()
    assert float(diff.max()) <= 1e-3 and float(diff.mean()) <= 1e-6
    assert float(ff.abs().max()) > 0.3


def test_mask_propagation_bit_exact_full_size(env):
    from oracle import oracle as O

    ctx, _, _, g = env
    gen = torch.Generator(device="cuda").manual_seed(5)
    coarse = torch.randn((1, 3, 6, 10, 10), device="cuda", generator=gen)
    dvf = torch.nn.functional.interpolate(coarse, size=SHAPE, mode="trilinear", align_corners=True)[0].contiguous() * 4.0
    x = torch.arange(NX, device="cuda").view(1, 1, NX)
    y = torch.arange(NY, device="cuda").view(1, NY, 1)
    z = torch.arange(NZ, device="cuda").view(NZ, 1, 1)
    mask = (((x - 250) / 120.0) ** 2 + ((y - 260) / 100.0) ** 2 + ((z - 120) / 70.0) ** 2 < 1).to(torch.uint8).contiguous()
    out = torch.empty_like(mask)
    ctx.resample(mask, g, g, out, field=dvf, interp=_lib.INTERP_NEAREST, default_value=0, u8=True)
    torch.cuda.synchronize()
    origin = (-250.0, -250.0, 10.0)
    want = O.resample(O.Vol(mask.cpu().numpy(), SPACING, origin), O.Vol(mask.cpu().numpy(), SPACING, origin),
                      field_vol=O.Vol(dvf.cpu().numpy().astype(np.float64), SPACING, origin), interp=O.INTERP_NEAREST).arr
    assert np.array_equal(out.cpu().numpy(), want)
    assert 0 < int(out.sum()) < out.numel()
