"""Parity of every C-ABI operation against the oracle (CPU restatement of the ITK filters).

Each test runs twice: `emu` (CPU suite: the unmodified kernel sources compiled against the
test-only HIP stand-in, so indexing / LDS / barrier logic is exercised without a GPU) and `gpu`
(-m gpu: the real libplatipy_hip.so on cuda:0).  Integer / mask results must be bit-exact;
fp32 results carry the tolerance written next to the assertion (the oracle keeps the reference's
fp64 field, the product stores fp32).
"""
import numpy as np
import pytest

from oracle import oracle as O
from platipy_amd import _lib
from tests.helpers import phantom, random_dvf, smooth_noise

FLT_MAX = float(np.finfo(np.float32).max)

# (nz, ny, nx), spacing(x,y,z), origin
GRIDS = [
    ((13, 22, 45), (0.9, 1.1, 2.5), (320.0, -52.0, 60.0)),   # nx % 4 != 0, one tile in x
    ((12, 20, 72), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)),        # nx % 4 == 0, two tiles in x and y
]


def geom_of(shape, spacing, origin):
    return _lib.make_geom((shape[2], shape[1], shape[0]), spacing, origin)


def size_of(shape):
    return (shape[2], shape[1], shape[0])


def test_gauss_taps_match_oracle_and_kats():
    lib = _lib.load(__import__("tests.emu.build_emu", fromlist=["build"]).build())
    # build-derived known answers: centre -> edge
    kats = {
        (1.0, 0.1): [0.4745589, 0.2118383, 0.0508822],
        (2.25, 0.1): [0.3161258, 0.2323020, 0.1096351],
        (0.36, 0.1): [0.7383936, 0.1308032],
        (1.0, 0.01): [0.4668012, 0.2083754, 0.0500505, 0.0081735],
        (4.0, 0.01): [0.2085927, 0.1801245, 0.1185304, 0.0615941, 0.0261393, 0.0093154],
    }
    for (var, err), want in kats.items():
        got = np.array(_lib.gauss_taps(var, err, 32, lib=lib))
        r = (len(got) - 1) // 2
        assert r == len(want) - 1
        np.testing.assert_allclose(got[r:], want, atol=5e-8)
        np.testing.assert_array_equal(got, O.gaussian_operator(var, err, 32))  # same recipe -> same doubles
    # width cap: the half kernel may hold max_width + 1 coefficients
    got = _lib.gauss_taps(100.0, 0.001, 5, lib=lib)
    assert len(got) == 2 * 5 + 1
    np.testing.assert_array_equal(np.array(got), O.gaussian_operator(100.0, 0.001, 5))


@pytest.mark.parametrize("grid", GRIDS)
def test_discrete_gaussian(backend, grid):
    shape, spacing, origin = grid
    img = phantom(shape, seed=3)
    for var, mkw in [((4.0, 4.0, 4.0), 32), ((1.0, 2.0, 9.0), 8)]:
        want = O.discrete_gaussian(O.Vol(img, spacing, origin), var, mkw).arr
        src = backend.dev(img)
        dst = backend.empty(shape)
        backend.ctx.discrete_gaussian(src, dst, size_of(shape), spacing, var, 0.01, mkw, True)
        got = backend.host(dst)
        # fp32 accumulation of <= 65 taps of |v| <= 1000: a few ulp of 1000
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-3)


@pytest.mark.parametrize("shape", [(9, 14, 24), (7, 11, 18), (20, 37, 260)])
@pytest.mark.parametrize("variance", [0.5, 2.25, 9.0, 30.0, 64.0])
def test_fir_register_window_and_shuffle_passes_equal_the_legacy_pass(backend, shape, variance, monkeypatch):
    """The marching (register-window) y / z passes and the wavefront-shuffle x pass add the same taps in the same order as
    the one-output-per-thread pass they replace (zero taps pad the radius to its bucket): bit-identical volumes, for
    radii 1..16, rows that are / are not a multiple of four voxels, and rows longer than one wavefront's span."""
    img = phantom(shape, seed=12)
    size = (shape[2], shape[1], shape[0])
    out = {}
    for legacy in ("1", None):
        if legacy:
            monkeypatch.setenv("PP_FIR_LEGACY", legacy)
        else:
            monkeypatch.delenv("PP_FIR_LEGACY", raising=False)
        dst = backend.empty(shape)
        backend.ctx.discrete_gaussian(backend.dev(img), dst, size, (1.0, 1.0, 1.0), (variance,) * 3, 0.01, 64, True)
        out[legacy] = backend.host(dst).copy()
    np.testing.assert_array_equal(out["1"], out[None])
    f = np.stack([phantom(shape, seed=20 + c) for c in range(3)]) * np.float32(0.01)
    res = {}
    for legacy in ("1", None):
        if legacy:
            monkeypatch.setenv("PP_FIR_LEGACY", legacy)
        else:
            monkeypatch.delenv("PP_FIR_LEGACY", raising=False)
        d = backend.dev(f)
        backend.ctx.smooth_field(d, size, [np.sqrt(variance)] * 3)
        res[legacy] = backend.host(d).copy()
    np.testing.assert_array_equal(res["1"], res[None])


@pytest.mark.parametrize("shape", [(19, 26, 40), (33, 41, 48), (12, 70, 300)])
@pytest.mark.parametrize("variance", [1.0, 16.0, 64.0, 110.0])
def test_fir_sparse_passes_equal_the_one_output_per_thread_passes(backend, shape, variance, monkeypatch):
    """The pyramid's blur is produced only on the rows its resample reads (pp_discrete_gaussian_rows_f32).  The passes
    behind it (k_fir_march_sp: a register window with loads in flight K steps ahead, outputs formed only where the list
    names them; k_fir_x_row: an extended row in LDS; radius up to 24 / 32) form every output with the dense filter's own
    sequence of fmas: bit-identical, on the listed rows, to the one-output-per-thread kernels (PP_FIR_MARCH_SP=0) and to
    the dense filter, for lists with gaps, first / last rows, and radii beyond the axis length."""
    img = phantom(shape, seed=14)
    size = (shape[2], shape[1], shape[0])
    rng = np.random.default_rng(int(variance) + shape[0])
    need_y = (rng.random(shape[1]) < 0.3).astype(np.uint8)
    need_z = (rng.random(shape[0]) < 0.4).astype(np.uint8)
    need_y[[0, -1]] = 1
    need_z[[0, -1]] = 1
    out = {}
    for sp in ("0", "1"):
        monkeypatch.setenv("PP_FIR_MARCH_SP", sp)
        dst = backend.empty(shape)
        backend.ctx.discrete_gaussian_rows(backend.dev(img), dst, size, (1.0, 1.0, 1.0), (variance,) * 3, backend.dev(need_y), backend.dev(need_z),
                                           0.01, 64, True)
        out[sp] = backend.host(dst)[need_z.astype(bool)][:, need_y.astype(bool)].copy()
    np.testing.assert_array_equal(out["0"], out["1"])
    dense = backend.empty(shape)
    backend.ctx.discrete_gaussian(backend.dev(img), dense, size, (1.0, 1.0, 1.0), (variance,) * 3, 0.01, 64, True)
    np.testing.assert_array_equal(backend.host(dense)[need_z.astype(bool)][:, need_y.astype(bool)], out["1"])


@pytest.mark.parametrize("grid", GRIDS)
def test_smooth_field(backend, grid):
    shape, spacing, _ = grid
    f = random_dvf(shape, spacing, seed=5)
    sig = [1.5 / s for s in spacing]
    want = O.smooth_field(f.astype(np.float64), sig)
    d = backend.dev(f)
    backend.ctx.smooth_field(d, size_of(shape), sig, 0.1, 30)
    np.testing.assert_allclose(backend.host(d), want, rtol=0, atol=2e-6)


@pytest.mark.parametrize("grid", GRIDS)
def test_warp_same_grid(backend, grid):
    shape, spacing, origin = grid
    mov = phantom(shape, seed=7)
    f = random_dvf(shape, spacing, seed=11, max_mm=6.0)
    f[:, :2, :, :] *= 8.0  # push some voxels out of the buffer
    g = geom_of(shape, spacing, origin)
    want = O.warp_image(O.Vol(mov, spacing, origin), f.astype(np.float64), edge_value=FLT_MAX).arr
    out = backend.empty(shape)
    backend.ctx.warp(backend.dev(mov), backend.dev(f), g, FLT_MAX, out)
    got = backend.host(out)
    sent_w, sent_g = want == FLT_MAX, got == FLT_MAX
    # inside/outside decisions may differ only within fp rounding of the buffer edge
    assert (sent_w != sent_g).mean() < 1e-4
    both = ~(sent_w | sent_g)
    # fp32 lerp of |v| <= ~1100 with a fp32 displacement: |dM/dx| * 1e-6 voxel + a few ulp
    np.testing.assert_allclose(got[both], want[both], rtol=0, atol=5e-3)
    assert sent_w.sum() > 0


def test_warp_identity_and_integer_shift(backend):
    shape, spacing, origin = (9, 14, 24), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)
    mov = phantom(shape, seed=8)
    g = geom_of(shape, spacing, origin)
    out = backend.empty(shape)
    zero = np.zeros((3,) + shape, dtype=np.float32)
    backend.ctx.warp(backend.dev(mov), backend.dev(zero), g, FLT_MAX, out)
    np.testing.assert_array_equal(backend.host(out), mov)  # warp by zero is the identity, bit for bit
    sh = zero.copy()
    sh[0] = 2.0   # +2 voxels in x
    sh[1] = -1.0  # -1 voxel in y
    backend.ctx.warp(backend.dev(mov), backend.dev(sh), g, -7.0, out)
    got = backend.host(out)
    want = np.full(shape, -7.0, dtype=np.float32)
    want[:, 1:, :-2] = mov[:, :-1, 2:]
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("interp", [_lib.INTERP_LINEAR, _lib.INTERP_NEAREST])
def test_resample_between_grids(backend, interp):
    shape, spacing, origin = GRIDS[0]
    img = phantom(shape, seed=9)
    vin = O.Vol(img, spacing, origin)
    # shrink-by-2 pyramid level grid (registration/utils.py:245-255)
    new_size = [int(s / 2.0 + 0.5) for s in vin.size]
    new_spacing = [((so - 1) * sp) / (sn - 1) for so, sp, sn in zip(vin.size, spacing, new_size)]
    ref = O.Vol(np.zeros(new_size[::-1], dtype=np.float32), new_spacing, origin)
    want = O.resample(vin, ref, interp=interp, default_value=-3.0).arr
    out = backend.empty(ref.arr.shape)
    backend.ctx.resample(backend.dev(img), geom_of(shape, spacing, origin), _lib.make_geom(new_size, new_spacing, origin),
                         out, interp=interp, default_value=-3.0)
    got = backend.host(out)
    if interp == _lib.INTERP_NEAREST:
        np.testing.assert_array_equal(got, want)
    else:
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-3)


def test_resample_affine(backend):
    shape, spacing, origin = GRIDS[1]
    img = phantom(shape, seed=10)
    vin = O.Vol(img, spacing, origin)
    ang = 0.1
    A = np.array([[np.cos(ang), -np.sin(ang), 0.02], [np.sin(ang), np.cos(ang), 0.0], [0.01, 0.0, 1.05]])
    t = np.array([1.5, -2.25, 0.5])
    ref_shape, ref_sp, ref_or = (10, 18, 40), (1.3, 1.1, 1.2), (2.0, 1.0, -1.0)
    ref = O.Vol(np.zeros(ref_shape, dtype=np.float32), ref_sp, ref_or)
    for interp, tol in [(_lib.INTERP_LINEAR, 2e-3), (_lib.INTERP_NEAREST, 0.0)]:
        want = O.resample(vin, ref, affine=(A, t), interp=interp, default_value=-1000.0).arr
        out = backend.empty(ref_shape)
        backend.ctx.resample(backend.dev(img), geom_of(shape, spacing, origin), geom_of(ref_shape, ref_sp, ref_or), out,
                             affine_A=A.ravel(), affine_t=t, interp=interp, default_value=-1000.0)
        got = backend.host(out)
        if tol == 0.0:
            np.testing.assert_array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=0, atol=tol)
        assert (want == -1000.0).any() and (want != -1000.0).any()


@pytest.mark.parametrize("grid", GRIDS)
def test_mask_propagation_bit_exact(backend, grid):
    """Propagated binary masks (nearest neighbour through a DVF) are bit-exact given the same DVF."""
    shape, spacing, origin = grid
    rng = np.random.default_rng(2)
    mask = (smooth_noise(shape, 21, cells=4) > 0.2).astype(np.uint8)
    mask[rng.integers(0, shape[0], 40), rng.integers(0, shape[1], 40), rng.integers(0, shape[2], 40)] ^= 1
    f = random_dvf(shape, spacing, seed=13, max_mm=7.0)
    vin = O.Vol(mask, spacing, origin)
    fvol = O.Vol(f.astype(np.float64), spacing, origin)
    want = O.resample(vin, vin, field_vol=fvol, interp=_lib.INTERP_NEAREST, default_value=0).arr
    out = backend.empty(shape, np.uint8)
    g = geom_of(shape, spacing, origin)
    backend.ctx.resample(backend.dev(mask), g, g, out, field=backend.dev(f), interp=_lib.INTERP_NEAREST, default_value=0,
                         u8=True)
    got = backend.host(out)
    assert got.dtype == np.uint8
    np.testing.assert_array_equal(got, want)
    assert 0 < want.sum() < want.size


def test_resample_field_and_compose(backend):
    shape, spacing, origin = GRIDS[0]
    f = random_dvf(shape, spacing, seed=17, max_mm=5.0)
    fin = O.Vol(f.astype(np.float64), spacing, origin)
    # up-sample onto a finer grid with the same corners (deformable.py:137)
    fine_shape = (shape[0] * 2 - 1, shape[1] * 2 - 1, shape[2] * 2 - 1)
    fine_sp = [sp * (n - 1) / (m - 1) for sp, n, m in zip(spacing, size_of(shape), size_of(fine_shape))]
    ref = O.Vol(np.zeros(fine_shape, dtype=np.float32), fine_sp, origin)
    want = O.resample_vec(fin, ref).arr
    out = backend.empty((3,) + fine_shape)
    backend.ctx.resample_field(backend.dev(f), geom_of(shape, spacing, origin), geom_of(fine_shape, fine_sp, origin), out)
    np.testing.assert_allclose(backend.host(out), want, rtol=0, atol=2e-5)
    # composition total += iter o (id + total)  (deformable.py:154)
    it = random_dvf(shape, spacing, seed=19, max_mm=3.0)
    tot = O.Vol(f.astype(np.float64), spacing, origin)
    comp = O.resample_vec(O.Vol(it.astype(np.float64), spacing, origin), tot, through=tot).arr
    want2 = tot.arr + comp
    d_tot = backend.dev(f)
    backend.ctx.compose_field(d_tot, backend.dev(it), geom_of(shape, spacing, origin))
    np.testing.assert_allclose(backend.host(d_tot), want2, rtol=0, atol=2e-5)


@pytest.mark.parametrize("shape", [(23, 30, 70), (37, 9, 129), (6, 5, 1), (40, 21, 2)])
def test_marching_gathers_equal_the_general_kernels(backend, shape, monkeypatch):
    """Round 5: on axis-aligned grids below 2^32 bytes, resample (linear / nearest, fp32 / u8, with and without a field) runs
    with one fp64 multiply per axis and 32-bit offsets, and the field up-sampling marches z with its corner pairs kept in
    registers from plane to plane; with a linear transform between the grids only the 3 x 3 in the middle stays.  The terms
    they drop are products with exact zeros, so every result is bit-identical to
    the general kernels (PP_RESAMPLE_GENERIC=1) -- inside, on the buffer border, outside, single-column and two-column
    volumes, tiles and z chunks that overhang the volume, up- and down-sampling along z."""
    spacing, origin = (0.9, 1.3, 2.1), (-11.5, 4.25, 100.0)
    rng = np.random.default_rng(77)
    img = (phantom(shape, seed=3) + 50.0 * rng.standard_normal(shape)).astype(np.float32)
    lab = (rng.integers(0, 5, shape)).astype(np.uint8)
    f = (random_dvf(shape, spacing, seed=5, max_mm=9.0) + rng.normal(size=(3,) + shape)).astype(np.float32)
    it = random_dvf(shape, spacing, seed=6, max_mm=4.0)
    g = geom_of(shape, spacing, origin)
    other_shape = (max(1, shape[0] * 2 - 1), max(1, shape[1] + 3), max(1, shape[2] // 2 + 1))
    other = geom_of(other_shape, (0.47, 1.21, 4.0), (-12.0, 3.0, 99.0))
    res = {}
    for mode in ("", "1"):
        if mode:
            monkeypatch.setenv("PP_RESAMPLE_GENERIC", mode)
        else:
            monkeypatch.delenv("PP_RESAMPLE_GENERIC", raising=False)
        ctx, r = backend.ctx, []
        for interp in (_lib.INTERP_LINEAR, _lib.INTERP_NEAREST):
            out = backend.empty(shape)
            ctx.resample(backend.dev(img), g, g, out, field=backend.dev(f), interp=interp, default_value=-1000.0)
            r.append(backend.host(out).copy())
            out = backend.empty(other_shape)
            ctx.resample(backend.dev(img), g, other, out, interp=interp, default_value=-3.0)
            r.append(backend.host(out).copy())
            out = backend.empty(shape, np.uint8)
            ctx.resample(backend.dev(lab), g, g, out, field=backend.dev(f), interp=interp, default_value=7, u8=True)
            r.append(backend.host(out).copy())
            out = backend.empty(other_shape, np.uint8)
            ctx.resample(backend.dev(lab), g, other, out, interp=interp, default_value=0, u8=True)
            r.append(backend.host(out).copy())
        # the same with a linear transform between the two grids (the pipelines' affine propagation)
        ang = 0.07
        A = np.array([[np.cos(ang), -np.sin(ang), 0.01], [np.sin(ang), np.cos(ang), -0.02], [0.015, 0.0, 1.03]])
        t = np.array([0.8, -1.1, 0.6])
        for interp in (_lib.INTERP_LINEAR, _lib.INTERP_NEAREST):
            out = backend.empty(other_shape)
            ctx.resample(backend.dev(img), g, other, out, affine_A=A.ravel(), affine_t=t, interp=interp, default_value=-1000.0)
            r.append(backend.host(out).copy())
            out = backend.empty(shape)
            ctx.resample(backend.dev(img), g, g, out, affine_A=A.ravel(), affine_t=t, field=backend.dev(f), interp=interp, default_value=-2.0)
            r.append(backend.host(out).copy())
            out = backend.empty(other_shape, np.uint8)
            ctx.resample(backend.dev(lab), g, other, out, affine_A=A.ravel(), affine_t=t, interp=interp, default_value=9, u8=True)
            r.append(backend.host(out).copy())
        out = backend.empty((3,) + other_shape)
        ctx.resample_field(backend.dev(f), g, other, out)
        r.append(backend.host(out).copy())
        fine_shape = (shape[0] * 3 + 2, shape[1] * 2, shape[2] + 1)       # x3 along z: most steps reuse both planes
        fine = geom_of(fine_shape, (0.9 * shape[2] / (shape[2] + 1), 0.66, 0.7), (-11.9, 4.0, 99.0))
        out = backend.empty((3,) + fine_shape)
        ctx.resample_field(backend.dev(f), g, fine, out)
        r.append(backend.host(out).copy())
        tot = backend.dev(f)
        ctx.compose_field(tot, backend.dev(it), g)
        r.append(backend.host(tot).copy())
        res[mode] = r
    for a, b in zip(res[""], res["1"]):
        np.testing.assert_array_equal(a, b)
    assert any((a != a.flat[0]).any() for a in res[""])


@pytest.mark.parametrize("shape", [(3, 260, 9), (2, 65, 300)])
def test_banded_launch_visits_every_voxel_once(backend, shape, monkeypatch):
    """Round 5: the gathers through a field (warp, axis-aligned resample, composition) are launched as a 1-D grid whose
    block b works on plane (b / 8) / per, tile (b % 8) x per + (b / 8) % per, so that each XCD owns a band of rows of every
    plane (PP_RS_BAND=0: the plain 3-D grid).  Same results, on grids whose tile count is not a multiple of 8."""
    spacing, origin = (1.1, 0.8, 2.0), (3.0, -2.0, 1.0)
    rng = np.random.default_rng(5)
    img = (phantom(shape, seed=8) + 20.0 * rng.standard_normal(shape)).astype(np.float32)
    lab = rng.integers(0, 4, shape).astype(np.uint8)
    f = random_dvf(shape, spacing, seed=9, max_mm=3.0).astype(np.float32)
    it = random_dvf(shape, spacing, seed=10, max_mm=2.0)
    g = geom_of(shape, spacing, origin)
    res = {}
    for mode in ("", "0"):
        if mode:
            monkeypatch.setenv("PP_RS_BAND", mode)
        else:
            monkeypatch.delenv("PP_RS_BAND", raising=False)
        ctx, r = backend.ctx, []
        out = backend.empty(shape)
        out[...] = -5.0
        ctx.warp(backend.dev(img), backend.dev(f), g, -1000.0, out)
        r.append(backend.host(out).copy())
        out = backend.empty(shape)
        out[...] = -5.0
        ctx.resample(backend.dev(img), g, g, out, field=backend.dev(f), interp=_lib.INTERP_LINEAR, default_value=-1000.0)
        r.append(backend.host(out).copy())
        out = backend.empty(shape, np.uint8)
        out[...] = 9
        ctx.resample(backend.dev(lab), g, g, out, field=backend.dev(f), interp=_lib.INTERP_NEAREST, default_value=7, u8=True)
        r.append(backend.host(out).copy())
        tot = backend.dev(f)
        ctx.compose_field(tot, backend.dev(it), g)
        r.append(backend.host(tot).copy())
        res[mode] = r
    for a, b in zip(res[""], res["0"]):
        np.testing.assert_array_equal(a, b)
    assert (res[""][0] != -5.0).all() and (res[""][1] != -5.0).all() and (res[""][2] != 9).all()
    fvol = O.Vol(f.astype(np.float64), spacing, origin)
    want = O.resample(O.Vol(lab, spacing, origin), O.Vol(lab, spacing, origin), field_vol=fvol, interp=_lib.INTERP_NEAREST, default_value=7).arr
    np.testing.assert_array_equal(res[""][2], want)


def _demons_params(ctx, iterations, spacing, variant, max_rms=0.02):
    p = ctx.default_demons_params()
    p.iterations = iterations
    p.smooth_update = 1
    p.smooth_displacement = 1
    p.sigma_d_vox[:] = [1.5 / s for s in spacing]
    p.max_rms_error = max_rms
    p.variant = variant
    return p


@pytest.mark.parametrize("grid", GRIDS)
def test_demons_force_single_sweep(backend, grid):
    shape, spacing, origin = grid
    fix = phantom(shape, seed=30)
    mov = phantom(shape, seed=30, noise=0) + 40.0 * smooth_noise(shape, 31).astype(np.float32)
    warped = mov.copy()
    warped[:, :, :3] = FLT_MAX          # crunched border
    warped[4:6, 5:9, 10:14] = FLT_MAX   # interior hole -> one-sided differences around it
    warped[2, 3, 20] = fix[2, 3, 20]    # |speed| below the intensity threshold
    p = _demons_params(backend.ctx, 1, spacing, _lib.DEMONS_STAGED)
    want, wst = O.esm_update(O.Vol(fix, spacing, origin), O.Vol(warped, spacing, origin))
    upd = backend.empty((3,) + shape)
    st = backend.ctx.demons_force(backend.dev(fix), backend.dev(warped), geom_of(shape, spacing, origin), p, upd)
    got = backend.host(upd)
    # |U| <= 0.5 * spacing; fp32 evaluation of 2 s J / (|J|^2 + s^2/K)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
    assert st.n_pixels == wst.n_pixels
    np.testing.assert_allclose(st.metric, wst.metric, rtol=1e-6)
    np.testing.assert_allclose(st.rms_change, wst.rms_change, rtol=1e-5)
    assert (got[:, 4:6, 5:9, 10:14] == 0).all()


def _oracle_execute(fix, mov, spacing, origin, iterations, max_rms):
    f = O.DemonsFilter()
    f.SetSmoothUpdateField(True)
    f.SetSmoothDisplacementField(True)
    f.SetStandardDeviations([1.5 / s for s in spacing])
    f.SetNumberOfIterations(iterations)
    f.SetMaximumRMSError(max_rms)
    d = f.Execute(O.Vol(fix, spacing, origin), O.Vol(mov, spacing, origin))
    return d.arr, f.stats


HIRES = ((14, 30, 70), (0.55, 0.62, 1.0), (1.0, 2.0, 3.0))   # sigma_d = 1.5 mm -> kernel radii 5, 4, 2


@pytest.mark.parametrize("variant", [_lib.DEMONS_STAGED, _lib.DEMONS_FUSED])
@pytest.mark.parametrize("grid", GRIDS + [HIRES])
def test_demons_execute(backend, grid, variant):
    """registration_algorithm.Execute (deformable.py:149): 4 iterations, field vs the fp64 oracle.

    Stated fp32 tolerance: max |D - D_oracle| <= 2e-3 mm (spacing ~1 mm) after 4 iterations on a
    noisy phantom; typical error is ~1e-5 mm."""
    shape, spacing, origin = grid
    fix = phantom(shape, seed=40)
    dv = random_dvf(shape, spacing, seed=41, max_mm=2.5)
    mov = O.warp_image(O.Vol(fix, spacing, origin), dv.astype(np.float64), edge_value=-1000.0).arr
    mov = (mov + np.random.default_rng(42).normal(0, 5, size=shape)).astype(np.float32)
    want, wst = _oracle_execute(fix, mov, spacing, origin, 4, 0.0)
    p = _demons_params(backend.ctx, 4, spacing, variant, max_rms=0.0)
    field = backend.empty((3,) + shape)
    st = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, field)
    got = backend.host(field)
    assert st.elapsed_iterations == 4 == wst.elapsed_iterations
    err = np.abs(got - want)
    assert err.max() <= 2e-3, f"max field error {err.max()} mm"
    assert np.sqrt((err ** 2).mean()) <= 5e-5
    np.testing.assert_allclose(st.metric, wst.metric, rtol=1e-4)
    np.testing.assert_allclose(st.rms_change, wst.rms_change, rtol=1e-4)
    assert np.abs(want).max() > 0.2  # the registration did something


MIXED = [((8, 37, 85), (1.5, 1.5, 1.5), (0.0, 0.0, 0.0)),      # 85 = 64 + 21: one column of 64 x 16 tiles, one of 32 x 32 (odd rows)
         ((7, 35, 160), (1.0, 1.2, 0.9), (5.0, 0.0, -3.0))]    # 160 = 2 x 64 + 32: two columns + a full 32-wide one (even rows: MASK)


@pytest.mark.parametrize("grid", GRIDS + [HIRES] + MIXED)
def test_fused_demons_tile_shapes_agree(backend, grid, monkeypatch):
    """The 32 x 32 tile variant of the fused kernels (chosen for grids that 64 x 16 tiles fit badly) computes every
    output voxel with the same operations in the same order: bit-identical fields, equal statistics.  So does the MIXED
    launch (64 x 16 tiles where a whole one fits, one column of 32 x 32 tiles over the rest), which the launcher picks by
    itself for row lengths 64 k + 1 .. 64 k + 32 of volumes from 8 M voxels up when no shape is forced."""
    shape, spacing, origin = grid
    fix = phantom(shape, seed=40)
    dv = random_dvf(shape, spacing, seed=41, max_mm=2.5)
    mov = O.warp_image(O.Vol(fix, spacing, origin), dv.astype(np.float64), edge_value=-1000.0).arr.astype(np.float32)
    p = _demons_params(backend.ctx, 3, spacing, _lib.DEMONS_FUSED, max_rms=0.0)
    out = {}
    for tile in ("0", "1", "mixed", "never mixed"):
        monkeypatch.delenv("PP_FUSED_TILE", raising=False)
        monkeypatch.delenv("PP_FUSED_MIX", raising=False)
        if tile in ("0", "1"):
            monkeypatch.setenv("PP_FUSED_TILE", tile)
        else:       # (on its own the launcher mixes from 8 M voxels up)
            monkeypatch.setenv("PP_FUSED_MIX", "1" if tile == "mixed" else "0")
        f = backend.empty((3,) + shape)
        st = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, f)
        out[tile] = (backend.host(f).copy(), st.metric, st.rms_change, st.elapsed_iterations)
    for other in ("1", "mixed", "never mixed"):
        np.testing.assert_array_equal(out["0"][0].view(np.uint32), out[other][0].view(np.uint32))
        np.testing.assert_allclose(out["0"][1:3], out[other][1:3], rtol=1e-6)
        assert out["0"][3] == out[other][3] == 3
    assert np.abs(out["0"][0]).max() > 0.1


ODD = ((9, 21, 67), (1.3, 0.9, 0.8), (0.0, 0.0, 0.0))   # odd nx: the scalar-store path; radii 1, 2, 2


@pytest.mark.parametrize("grid", [GRIDS[0], ODD, MIXED[0], ((6, 18, 70), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0))])
def test_fused_demons_padded_rows_do_not_change_the_field(backend, grid, monkeypatch):
    """Rows that are not whole 16-byte quads (nx % 4 != 0) are copied into padded rows for the generation-2 kernels (aligned
    strips and pairs, the MASK instances for odd row lengths too): the field, the statistics and the iteration count are
    those of the dense-row run, bit for bit, whatever the padding holds (the workspace is poisoned first)."""
    shape, spacing, origin = grid
    assert shape[2] % 4 != 0
    fix = phantom(shape, seed=60)
    dv = random_dvf(shape, spacing, seed=61, max_mm=2.5)
    mov = O.warp_image(O.Vol(fix, spacing, origin), dv.astype(np.float64), edge_value=-1000.0).arr.astype(np.float32)
    out = {}
    for iters in (3, 4):      # the newest field ends in either buffer
        p = _demons_params(backend.ctx, iters, spacing, _lib.DEMONS_FUSED, max_rms=0.0)
        for pitch, mask in (("1", "1"), ("1", "0"), ("0", "0")):     # (padded rows make the MASK instances possible for odd rows)
            monkeypatch.setenv("PP_FUSED_PITCH", pitch)
            monkeypatch.setenv("PP_FUSED_MASK", mask)
            f = backend.empty((3,) + shape)
            st = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, f)
            out[pitch + mask] = (backend.host(f).copy(), st.metric, st.rms_change, st.elapsed_iterations, st.n_pixels)
        for leg in ("11", "10"):
            np.testing.assert_array_equal(out[leg][0].view(np.uint32), out["00"][0].view(np.uint32))
            assert out[leg][1:] == out["00"][1:] and out[leg][3] == iters
        assert np.abs(out["00"][0]).max() > 0.1


@pytest.mark.parametrize("tile", ["0", "1"])
@pytest.mark.parametrize("grid", GRIDS + [HIRES, ODD])
def test_fused_demons_generations_agree(backend, grid, tile, monkeypatch):
    """The second-generation fused kernels (buffer addressing, two barrier intervals per plane, renamed z window,
    straight-line warp) perform the first generation's arithmetic operation for operation: bit-identical fields,
    and warped images (through the next iteration), equal statistics, for both tile shapes, radii 1..5 and odd row lengths."""
    shape, spacing, origin = grid
    fix = phantom(shape, seed=40)
    dv = random_dvf(shape, spacing, seed=41, max_mm=2.5)
    mov = O.warp_image(O.Vol(fix, spacing, origin), dv.astype(np.float64), edge_value=-1000.0).arr.astype(np.float32)
    p = _demons_params(backend.ctx, 3, spacing, _lib.DEMONS_FUSED, max_rms=0.0)
    monkeypatch.setenv("PP_FUSED_TILE", tile)
    out = {}
    for gen in ("1", "2", "2-separate"):   # "2": kernel A stores D + U (one halo'd volume for kernel B); "2-separate": U alone
        monkeypatch.setenv("PP_FUSED_GEN", gen[0])
        monkeypatch.setenv("PP_FUSED_SUM", "0" if gen.endswith("separate") else "1")
        f = backend.empty((3,) + shape)
        st = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, f)
        out[gen] = (backend.host(f).copy(), st.metric, st.rms_change, st.n_pixels, st.elapsed_iterations)
    np.testing.assert_array_equal(out["1"][0], out["2"][0])
    np.testing.assert_array_equal(out["2"][0], out["2-separate"][0])
    assert out["2"][1:] == out["2-separate"][1:]
    # the statistics are fp32 per-thread partial sums folded in fp64: the two generations group voxels differently
    np.testing.assert_allclose(out["1"][1:3], out["2"][1:3], rtol=1e-6)
    assert out["1"][3] == out["2"][3] and out["1"][4] == out["2"][4] == 3
    assert np.abs(out["2"][0]).max() > 0.1


BRICKS = [GRIDS[1], ((9, 21, 67), (1.0, 1.1, 1.2), (0.0, 0.0, 0.0)), MIXED[0],   # (radii <= 2; odd rows: scalar stores)
          ((20, 31, 38), (1.0, 1.2, 1.1), (5.0, 0.0, -3.0)),     # 4 x 4 x 3 bricks, every axis overhung
          ((7, 8, 16), (1.1, 1.1, 1.1), (0.0, 0.0, 0.0)),        # one brick in x and y: every halo slot is a clamped one
          ((15, 19, 30), (0.85, 0.9, 2.0), (0.0, 0.0, 0.0)),     # sigma_d 1.76 / 1.67 / 0.75 voxels: field radius 3, 3, 1
          ((5, 3, 4), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0))]         # smaller than the halo


@pytest.mark.parametrize("grid", BRICKS)
def test_fused_demons_bricks_equal_the_marching_kernels(backend, grid, monkeypatch):
    """Round 6: the latency-bound pyramid levels run generation 3 (pp_demons_cube.h: a block owns a 16 x 8 x 6 brick with its
    whole halo in LDS instead of marching a tile through z).  Same arithmetic voxel by voxel: the field -- and through the
    following iterations the warped image, the sentinel voxels and the clamped halo -- equals generation 2's bit for bit,
    with odd rows, overhung bricks and the RMS halt live; the statistics agree as sums grouped differently do."""
    shape, spacing, origin = grid
    fix = phantom(shape, seed=40)
    dv = random_dvf(shape, spacing, seed=41, max_mm=2.5)
    mov = O.warp_image(O.Vol(fix, spacing, origin), dv.astype(np.float64), edge_value=-1000.0).arr.astype(np.float32)
    backend.ctx.profile_enable(True)
    try:
        rms4 = None
        for iters, max_rms in ((3, 0.0), (4, 0.0), (12, None)):   # (the last leg halts early: a threshold just above iteration 4's RMS)
            p = _demons_params(backend.ctx, iters, spacing, _lib.DEMONS_FUSED, max_rms=1.02 * rms4 if max_rms is None else max_rms)
            out = {}
            for cube in ("0", "1"):
                monkeypatch.setenv("PP_FUSED_CUBE", cube)
                backend.ctx.profile_read()
                f = backend.empty((3,) + shape)
                st = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, f)
                names = {k for k, v in backend.ctx.profile_read().items() if v[0] > 0}
                assert ("k_cube_force_smooth" in names) == (cube == "1") and ("k_fused2_force_smooth" in names) == (cube == "0"), names
                out[cube] = (backend.host(f).copy(), st.metric, st.rms_change, st.n_pixels, st.elapsed_iterations, st.halted)
            np.testing.assert_array_equal(out["0"][0].view(np.uint32), out["1"][0].view(np.uint32))
            np.testing.assert_allclose(out["0"][1:3], out["1"][1:3], rtol=1e-6)
            assert out["0"][3:] == out["1"][3:]
            assert np.abs(out["1"][0]).max() > 0.05
            if iters == 4:
                rms4 = out["1"][2]
            if iters == 12:
                assert out["1"][5] == 1 and out["1"][4] < 12, out["1"][1:]
    finally:
        backend.ctx.profile_enable(False)
    monkeypatch.delenv("PP_FUSED_CUBE")
    f = backend.empty((3,) + shape)       # and the launcher picks the bricks by itself at these sizes
    backend.ctx.profile_enable(True)
    backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, f)
    names = {k for k, v in backend.ctx.profile_read().items() if v[0] > 0}
    backend.ctx.profile_enable(False)
    assert "k_cube_force_smooth" in names and "k_cube_smooth_warp" in names and "k_fused2_force_smooth" not in names, names
    np.testing.assert_array_equal(backend.host(f).view(np.uint32), out["1"][0].view(np.uint32))


def test_fused_demons_store_policy_does_not_change_the_field(backend, monkeypatch):
    """Generation 2 stores its outputs with a non-temporal hint on volumes far beyond the infinity cache (a template
    parameter chosen per launch); forcing either policy on a small grid gives the same bits, statistics included."""
    shape, spacing, origin = GRIDS[0]
    fix = phantom(shape, seed=40)
    dv = random_dvf(shape, spacing, seed=41, max_mm=2.5)
    mov = O.warp_image(O.Vol(fix, spacing, origin), dv.astype(np.float64), edge_value=-1000.0).arr.astype(np.float32)
    p = _demons_params(backend.ctx, 3, spacing, _lib.DEMONS_FUSED, max_rms=0.0)
    out = {}
    for nt in ("0", "1"):
        monkeypatch.setenv("PP_FUSED_NT", nt)
        f = backend.empty((3,) + shape)
        st = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, f)
        out[nt] = (backend.host(f).copy(), st.metric, st.rms_change, st.n_pixels, st.elapsed_iterations)
    np.testing.assert_array_equal(out["0"][0], out["1"][0])
    assert out["0"][1:] == out["1"][1:]


EVEN_ROWS = [GRIDS[1], HIRES, ((13, 22, 46), (0.9, 1.1, 2.5), (320.0, -52.0, 60.0))]   # nx % 4 == 0 and nx % 4 == 2


@pytest.mark.parametrize("tile", ["0", "1"])
@pytest.mark.parametrize("grid", EVEN_ROWS)
def test_fused_demons_masked_kernels_equal_the_branchy_ones(backend, grid, tile, monkeypatch):
    """Round 4: on grids with even rows generation 2 runs its MASK instances -- every memory instruction of a plane step
    issued on every step, loads from clamped addresses, the lane mask of a store carried by an out-of-range buffer offset
    -- so that the compiler's s_waitcnt pass sees straight-line code.  Same arithmetic: the field, the warped image it
    feeds the next iteration and the statistics equal those of the branchy instances (PP_FUSED_MASK=0) bit for bit, on
    grids whose tiles overhang the volume in x and y, with z-chunks shorter than the halo included."""
    shape, spacing, origin = grid
    assert shape[2] % 2 == 0, "the masked instances need even rows"
    fix = phantom(shape, seed=40)
    dv = random_dvf(shape, spacing, seed=41, max_mm=2.5)
    mov = O.warp_image(O.Vol(fix, spacing, origin), dv.astype(np.float64), edge_value=-1000.0).arr.astype(np.float32)
    p = _demons_params(backend.ctx, 3, spacing, _lib.DEMONS_FUSED, max_rms=0.0)
    monkeypatch.setenv("PP_FUSED_TILE", tile)
    for zchunk in (None, "2"):
        if zchunk is None:
            monkeypatch.delenv("PP_FUSED_ZCHUNK", raising=False)
        else:
            monkeypatch.setenv("PP_FUSED_ZCHUNK", zchunk)
        out = {}
        for mask in ("0", "1"):
            monkeypatch.setenv("PP_FUSED_MASK", mask)
            f = backend.empty((3,) + shape)
            st = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, f)
            out[mask] = (backend.host(f).copy(), st.metric, st.rms_change, st.n_pixels, st.elapsed_iterations)
        np.testing.assert_array_equal(out["0"][0], out["1"][0])
        assert out["0"][1:] == out["1"][1:]
        assert np.abs(out["1"][0]).max() > 0.1


@pytest.mark.parametrize("zchunk", [1, 2, 3, 5, 100])
def test_fused_demons_is_independent_of_the_z_chunking(backend, zchunk, monkeypatch):
    """The fused schedule splits z into chunks (halo planes recomputed at the seams); any chunk length, shorter
    than the halo included, gives bit-identical fields."""
    shape, spacing, origin = GRIDS[0]
    fix = phantom(shape, seed=40)
    dv = random_dvf(shape, spacing, seed=41, max_mm=2.5)
    mov = O.warp_image(O.Vol(fix, spacing, origin), dv.astype(np.float64), edge_value=-1000.0).arr.astype(np.float32)
    p = _demons_params(backend.ctx, 3, spacing, _lib.DEMONS_FUSED, max_rms=0.0)
    monkeypatch.delenv("PP_FUSED_ZCHUNK", raising=False)
    ref = backend.empty((3,) + shape)
    st0 = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, ref)
    monkeypatch.setenv("PP_FUSED_ZCHUNK", str(zchunk))
    out = backend.empty((3,) + shape)
    st1 = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, out)
    np.testing.assert_array_equal(backend.host(out), backend.host(ref))
    # the statistics are sums of per-block partial sums: their grouping follows the chunking
    np.testing.assert_allclose([st1.metric, st1.rms_change], [st0.metric, st0.rms_change], rtol=1e-6)


@pytest.mark.parametrize("variant", [_lib.DEMONS_STAGED, _lib.DEMONS_FUSED])
def test_demons_early_halt(backend, variant):
    """MaximumRMSError stops the loop on device exactly where FiniteDifferenceImageFilter::Halt does."""
    shape, spacing, origin = GRIDS[1]
    fix = phantom(shape, seed=50, noise=0)
    mov = fix + np.float32(0.5) * smooth_noise(shape, 51).astype(np.float32)
    # find an RMS threshold that the oracle crosses mid-run
    rms = []
    for n in range(1, 6):
        _, s = _oracle_execute(fix, mov, spacing, origin, n, 0.0)
        rms.append(s.rms_change)
    thr = 0.5 * (rms[1] + rms[2])
    want, wst = _oracle_execute(fix, mov, spacing, origin, 8, thr)
    assert 1 < wst.elapsed_iterations < 8
    p = _demons_params(backend.ctx, 8, spacing, variant, max_rms=thr)
    field = backend.empty((3,) + shape)
    st = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, field)
    assert st.elapsed_iterations == wst.elapsed_iterations
    assert st.halted == 1
    np.testing.assert_allclose(backend.host(field), want, rtol=0, atol=1e-3)  # fp32 field vs fp64 oracle


@pytest.mark.parametrize("grid", GRIDS)
def test_recursive_gaussian_field(backend, grid):
    shape, spacing, origin = grid
    f = random_dvf(shape, spacing, seed=60, max_mm=5.0)
    for sigma in ([1.5 / s for s in spacing], [0.4, 2.0, 3.0]):
        want = O.recursive_gaussian_vec(O.Vol(f.astype(np.float64), spacing, origin), sigma).arr
        d = backend.dev(f)
        backend.ctx.recursive_gaussian_field(d, geom_of(shape, spacing, origin), sigma)
        # each directional pass stores fp32; the causal half is rounded once more than in ITK
        np.testing.assert_allclose(backend.host(d), want, rtol=0, atol=3e-6)


def test_recursive_gaussian_needs_four_voxels(backend):
    shape = (3, 8, 8)
    d = backend.dev(np.zeros((3,) + shape, dtype=np.float32))
    with pytest.raises(_lib.PlatipyAmdError):
        backend.ctx.recursive_gaussian_field(d, geom_of(shape, (1, 1, 1), (0, 0, 0)), [1.0, 1.0, 1.0])


@pytest.mark.parametrize("grid", GRIDS)
def test_weight_map_and_fusion_ops(backend, grid):
    shape, spacing, origin = grid
    n = int(np.prod(shape))
    tgt = phantom(shape, seed=70)
    atl = [phantom(shape, seed=70, noise=0) + 30 * smooth_noise(shape, 71 + i).astype(np.float32) for i in range(3)]
    labels = [(smooth_noise(shape, 80 + i, cells=4) > 0.1).astype(np.uint8) for i in range(3)]
    ctx = backend.ctx
    wsum = backend.empty(shape)
    wlsum = backend.empty(shape)
    aset = {}
    for i in range(3):
        want_w = O.compute_weight_map(O.Vol(tgt, spacing, origin), O.Vol(atl[i], spacing, origin), "local").arr
        w = backend.empty(shape)
        ctx.weight_map_local(backend.dev(tgt), backend.dev(atl[i]), size_of(shape), spacing, 2.0, 1e-5, w)
        got_w = backend.host(w)
        np.testing.assert_allclose(got_w, want_w, rtol=3e-5, atol=0)
        # global vote: fp64 sum of squared differences
        ssd = ctx.sum_sq_diff(backend.dev(tgt), backend.dev(atl[i]), n)
        np.testing.assert_allclose(ssd, ((tgt.astype(np.float64) - atl[i]) ** 2).sum(), rtol=1e-12)
        # accumulate with the oracle's weights so the fold itself is compared exactly
        ctx.fuse_accumulate(backend.dev(want_w), backend.dev(labels[i]), wsum, wlsum, n)
        aset[str(i)] = {"DIR": {"Weight Map": O.Vol(want_w, spacing, origin), "S": O.Vol(labels[i], spacing, origin)}}
    ws = [aset[str(i)]["DIR"]["Weight Map"].arr for i in range(3)]
    np.testing.assert_array_equal(backend.host(wsum), (ws[0] + ws[1]) + ws[2])  # same left fold, same fp32 adds
    prob = backend.empty(shape)
    ctx.fuse_divide(wlsum, wsum, prob, n)
    ctx.discrete_gaussian(prob, prob, size_of(shape), spacing, (1.0, 1.0, 1.0), 0.01, 32, True)
    lo, hi = ctx.minmax(prob, n)
    ph = backend.host(prob)
    assert lo == ph.min() and hi == ph.max()
    ctx.rescale_threshold(prob, n, lo, hi, 1e-4)
    want_p = O.combine_labels(aset, "S")["S"].arr
    # w*L products may be contracted to fma on the GPU: <= 1 ulp of the running sum, then /, blur, rescale
    np.testing.assert_allclose(backend.host(prob), want_p, rtol=0, atol=3e-6)
    out = backend.empty(shape, np.uint8)
    lo2, hi2 = ctx.minmax(prob, n)
    ctx.binary_threshold(prob, n, hi2, 0.5, out)
    b = backend.host(out)
    pr = backend.host(prob)
    np.testing.assert_array_equal(b, ((pr.astype(np.float64) / hi2).astype(np.float32) >= 0.5).astype(np.uint8))


def test_fillhole_largest_component(backend):
    """BinaryFillhole -> ConnectedComponent -> largest (fusion.py:310-328) vs scipy, bit-exact."""
    from scipy import ndimage

    rng = np.random.default_rng(5)
    # (4, 9, 600): rows longer than one scan pass (run starts carried across passes); (44, 128, 160): more voxels than
    # 2048 blocks x 256, so that the size count's blocks walk several segments each (runs carried across segments)
    for shape, seed in [((12, 20, 37), 1), ((9, 33, 64), 2), ((6, 7, 5), 3), ((4, 9, 600), 4), ((44, 128, 160), 5)]:
        base = smooth_noise(shape, 90 + seed, cells=4) > 0.15
        m = base.copy()
        holes = rng.random(shape) < 0.04
        m &= ~holes                                   # punch holes (some open to the border, some enclosed)
        m |= rng.random(shape) < 0.01                 # specks: extra small components
        m = m.astype(np.uint8)
        for fill in (True, False):
            b = m.astype(bool)
            if fill:
                b = ndimage.binary_fill_holes(b)
            lab, ncomp = ndimage.label(b)
            counts = np.bincount(lab.ravel())[1:]
            want = (lab == 1 + int(np.argmax(counts))).astype(np.uint8)
            out = backend.empty(shape, np.uint8)
            cnt = backend.ctx.fillhole_largest_component(backend.dev(m), (shape[2], shape[1], shape[0]), out, fill_holes=fill,
                                                         want_count=True)
            np.testing.assert_array_equal(backend.host(out), want)
            assert cnt == counts.max()
    # ties: two equal components -> the first in raster order (np.argmax over ITK's label order)
    t = np.zeros((4, 6, 10), np.uint8)
    t[1, 1, 1:4] = 1
    t[2, 4, 5:8] = 1
    out = backend.empty(t.shape, np.uint8)
    backend.ctx.fillhole_largest_component(backend.dev(t), (10, 6, 4), out)
    want = np.zeros_like(t)
    want[1, 1, 1:4] = 1
    np.testing.assert_array_equal(backend.host(out), want)
    # empty mask: comes back unchanged
    z = np.zeros((4, 6, 10), np.uint8)
    backend.ctx.fillhole_largest_component(backend.dev(z), (10, 6, 4), out)
    assert backend.host(out).sum() == 0


@pytest.mark.parametrize("grid", GRIDS)
def test_distance_map_and_contour(backend, grid):
    """|SignedMaurerDistanceMap| and LabelContour (label/projection.py:80-90): exact EDT vs scipy."""
    shape, spacing, origin = grid
    mask = (smooth_noise(shape, 95, cells=4) > 0.2).astype(np.uint8)
    mask[:, :2, :] = 0
    mask[3:6, 8:14, 10:20] = 1
    mask[4, 10, 14] = 0  # interior hole -> extra border voxels
    g = geom_of(shape, spacing, origin)
    for signed in (False, True):
        want = O.maurer_distance_map(O.Vol(mask, spacing, origin), signed=signed).arr
        out = backend.empty(shape)
        backend.ctx.distance_map(backend.dev(mask), g, out, signed=signed)
        # exact EDT; fp32 squares of <= ~120 mm distances
        np.testing.assert_allclose(backend.host(out), want, rtol=2e-6, atol=2e-5)
    c = backend.empty(shape, np.uint8)
    backend.ctx.label_contour(backend.dev(mask), size_of(shape), c)
    np.testing.assert_array_equal(backend.host(c), O.label_contour(O.Vol(mask, spacing, origin)).arr)


def test_bounding_box(backend):
    """pp_bounding_box (label_to_roi, utils/crop.py:24-60): uint8 and float volumes, a single voxel, nothing at all."""
    shape = (9, 13, 70)                     # rows longer than a wave
    rng = np.random.default_rng(8)
    for dtype in (np.uint8, np.float32):
        a = np.zeros(shape, dtype)
        assert backend.ctx.bounding_box(backend.dev(a), size_of(shape), dtype == np.float32)[0] > backend.ctx.bounding_box(
            backend.dev(a), size_of(shape), dtype == np.float32)[1]
        a[4, 7, 66] = 3
        assert backend.ctx.bounding_box(backend.dev(a), size_of(shape), dtype == np.float32) == [66, 66, 7, 7, 4, 4]
        a[2:8, 3:11, 5:69] = (rng.random((6, 8, 64)) > 0.7).astype(dtype)
        if dtype == np.float32:
            a[0, 0, 0] = -5.0               # not > 0
        zz, yy, xx = np.nonzero(a > 0)
        want = [xx.min(), xx.max(), yy.min(), yy.max(), zz.min(), zz.max()]
        assert backend.ctx.bounding_box(backend.dev(a), size_of(shape), dtype == np.float32) == [int(v) for v in want]


@pytest.mark.parametrize("radius", [(1, 1, 1), (2, 2, 0), (3, 2, 1), (0, 0, 0), (5, 4, 2)])
def test_binary_morphology_ball(backend, radius):
    """BinaryDilate / BinaryErode / BinaryMorphologicalClosing with ITK's ball (registration/utils.py:328-329,
    multiatlas/run.py:421-423): bit-exact vs the oracle, objects touching the buffer edge included."""
    shape = (12, 20, 26)
    mask = (smooth_noise(shape, 311, cells=4) > 0.35).astype(np.uint8)
    mask[:3, :4, :5] = 1          # touches three faces
    mask[6, 10, 13] = 0
    vol = O.Vol(mask, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0))
    for op, ref in ((0, O.binary_dilate_ball), (1, O.binary_erode_ball), (2, O.binary_closing_ball)):
        out = backend.empty(shape, np.uint8)
        backend.ctx.binary_morph_ball(backend.dev(mask), size_of(shape), radius, op, out)
        np.testing.assert_array_equal(backend.host(out), ref(vol, radius).arr, err_msg=f"op {op}")


def test_ball_element_known_shapes():
    """ITK's radius-1 ball in 3-D is the 18-neighbourhood + centre (19 voxels); (2, 2, 0) is the 21-pixel disc."""
    assert int(O.ball_element((1, 1, 1)).sum()) == 19
    b = O.ball_element((2, 2, 0))
    assert b.shape == (1, 5, 5) and int(b.sum()) == 21 and not b[0, 0, 0]


@pytest.mark.parametrize("variant", [_lib.DEMONS_STAGED, _lib.DEMONS_FUSED])
def test_demons_edge_cases(backend, variant):
    """Degenerate inputs the reference's filter accepts: tiny volumes (smaller than one tile, shorter than the kernel
    radius), zero iterations, identical images (halts after the first iteration), a one-voxel-thick axis."""
    ctx = backend.ctx
    for shape in [(5, 6, 7), (1, 9, 11), (3, 1, 70)]:
        spacing, origin = (1.0, 1.3, 2.0), (0.0, 0.0, 0.0)
        rng = np.random.default_rng(sum(shape))
        fix = (rng.normal(size=shape) * 50).astype(np.float32)
        mov = (fix + rng.normal(size=shape) * 10).astype(np.float32)
        want, wst = _oracle_execute(fix, mov, spacing, origin, 3, 0.0)
        p = _demons_params(ctx, 3, spacing, variant, max_rms=0.0)
        field = backend.empty((3,) + shape)
        st = ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, field)
        assert st.elapsed_iterations == 3 and st.n_pixels == wst.n_pixels
        np.testing.assert_allclose(backend.host(field), want, rtol=0, atol=1e-4)
    shape, spacing, origin = (6, 10, 20), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)
    img = phantom(shape, seed=9)
    g = geom_of(shape, spacing, origin)
    # zero iterations: the field is zero and nothing was counted
    p = _demons_params(ctx, 0, spacing, variant)
    field = backend.dev(np.ones((3,) + shape, np.float32))
    st = ctx.demons_execute(backend.dev(img), backend.dev(img), g, p, field)
    assert st.elapsed_iterations == 0 and not backend.host(field).any()
    # identical images: zero update, metric 0, RMS change 0 < MaximumRMSError -> Halt() after one iteration
    p = _demons_params(ctx, 5, spacing, variant, max_rms=0.02)
    st = ctx.demons_execute(backend.dev(img), backend.dev(img), g, p, field)
    assert st.elapsed_iterations == 1 and st.halted == 1 and st.metric == 0.0 and st.rms_change == 0.0
    assert not backend.host(field).any()


def test_argument_errors(backend):
    ctx = backend.ctx
    shape = (4, 6, 8)
    a = backend.dev(np.zeros(shape, np.float32))
    g = geom_of(shape, (1, 1, 1), (0, 0, 0))
    p = ctx.default_demons_params()
    with pytest.raises(_lib.PlatipyAmdError):          # NULL volume
        ctx.demons_execute(a, None, g, p, backend.empty((3,) + shape))
    bad = _lib.make_geom((8, 6, 4), (1.0, 0.0, 1.0))   # zero spacing
    with pytest.raises(_lib.PlatipyAmdError):
        ctx.warp(a, backend.empty((3,) + shape), bad, 0.0, backend.empty(shape))
    with pytest.raises(_lib.PlatipyAmdError):          # in-place resample
        ctx.resample(a, g, g, a)
    p.variant = _lib.DEMONS_FUSED
    p.smooth_update = 0                                # the fused schedule needs both smoothers
    with pytest.raises(_lib.PlatipyAmdError):
        ctx.demons_execute(a, a, g, p, backend.empty((3,) + shape))
    rot = _lib.make_geom((8, 6, 4), (1, 1, 1), (0, 0, 0), (0, 1, 0, -1, 0, 0, 0, 0, 1))
    p = ctx.default_demons_params()
    with pytest.raises(_lib.PlatipyAmdError):          # direction cosines are handled above the ABI
        ctx.demons_execute(a, a, rot, p, backend.empty((3,) + shape))


WIDE = ((8, 40, 136), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0))   # 64 x 16 tiles: tile (1, 1) lies strictly inside the volume


@pytest.mark.parametrize("sentinels", [False, True])
def test_fused_kernels_on_interior_tiles(backend, sentinels, monkeypatch):
    """A grid with tiles strictly inside the volume (the other grids of this file only have border tiles, whose clamps and
    out-of-volume sentinel slots take different paths): generation 2 == generation 1 bit for bit, both == the oracle.
    `sentinels`: FLT_MAX voxels planted in the moving image inside the interior tile -- iteration 0 reads the moving image
    as the warped one, so ITK's sentinel case analysis is exercised away from the volume border too."""
    shape, spacing, origin = WIDE
    fix = phantom(shape, seed=50)
    dv = random_dvf(shape, spacing, seed=51, max_mm=2.0)
    mov = O.warp_image(O.Vol(fix, spacing, origin), dv.astype(np.float64), edge_value=-1000.0).arr.astype(np.float32)
    iters = 3
    if sentinels:
        mov[3, 20:23, 80:90] = np.finfo(np.float32).max
        mov[6, 25, 100] = np.finfo(np.float32).max
        iters = 1          # (later iterations would interpolate FLT_MAX into infinities: nothing to compare)
    monkeypatch.setenv("PP_FUSED_TILE", "0")
    p = _demons_params(backend.ctx, iters, spacing, _lib.DEMONS_FUSED, max_rms=0.0)
    out = {}
    for gen in ("2", "1"):
        monkeypatch.setenv("PP_FUSED_GEN", gen)
        f = backend.empty((3,) + shape)
        st = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, f)
        out[gen] = (backend.host(f).copy(), st.metric, st.rms_change, st.n_pixels)
    np.testing.assert_array_equal(out["2"][0].view(np.uint32), out["1"][0].view(np.uint32))
    np.testing.assert_allclose(out["2"][1:3], out["1"][1:3], rtol=1e-6)
    assert out["2"][3] == out["1"][3]
    want, wst = _oracle_execute(fix, mov, spacing, origin, iters, 0.0)
    err = np.abs(out["2"][0] - want)
    assert err.max() <= 2e-3 and np.sqrt((err ** 2).mean()) <= 5e-5
    assert out["2"][3] == wst.n_pixels
    np.testing.assert_allclose(out["2"][1], wst.metric, rtol=1e-4)


@pytest.mark.parametrize("variant", [_lib.DEMONS_STAGED, _lib.DEMONS_FUSED])
def test_demons_history_is_what_an_iteration_observer_reads(backend, variant):
    """pp_demons_history: entry k = GetMetric() / GetRMSChange() after iteration k + 1 (deformable.py:260-264,
    registration/utils.py:36-41), kept on the device by the kernel that closes each iteration.  Checked against runs that
    stop after k + 1 iterations, and with the early halt."""
    shape, spacing, origin = GRIDS[1]
    fix = phantom(shape, seed=40)
    dv = random_dvf(shape, spacing, seed=41, max_mm=2.5)
    mov = O.warp_image(O.Vol(fix, spacing, origin), dv.astype(np.float64), edge_value=-1000.0).arr.astype(np.float32)
    g = geom_of(shape, spacing, origin)
    p = _demons_params(backend.ctx, 5, spacing, variant, max_rms=0.0)
    f = backend.empty((3,) + shape)
    st = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), g, p, f)
    hist = backend.ctx.demons_history()
    assert len(hist) == st.elapsed_iterations == 5
    assert hist[-1] == (st.metric, st.rms_change)
    for k in (1, 3):
        pk = _demons_params(backend.ctx, k, spacing, variant, max_rms=0.0)
        sk = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), g, pk, f)
        assert hist[k - 1] == (sk.metric, sk.rms_change)
        assert len(backend.ctx.demons_history()) == k
    assert hist[0][0] > hist[-1][0]                 # the metric falls
    # early halt: the history ends where the loop did
    ph = _demons_params(backend.ctx, 5, spacing, variant, max_rms=0.5 * (hist[1][1] + hist[2][1]))
    sh = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), g, ph, f)
    hh = backend.ctx.demons_history()
    assert sh.halted and len(hh) == sh.elapsed_iterations == 3 and hh == hist[:3]


@pytest.mark.parametrize("shape", [(150, 71, 24), (12, 40, 150), (9, 33, 77)])
def test_recursive_gaussian_single_sweep_equals_two_sweeps(backend, monkeypatch, shape):
    """The single-sweep recursive Gaussian (segments of 32 voxels in registers, the anti-causal recursion warmed up over the
    following segment) against the exact two-sweep walk on lines longer than several segments, ragged in length: the
    warm-up error is < 1e-12 of the signal, so the fp32 results are equal except where a value sits within that distance
    of a rounding boundary (<= 1 ulp, a vanishing fraction); and both equal the oracle.  The second and third shapes have x
    rows of 150 and 77 voxels: rows that are not whole 16-byte quads (the x kernel's 4-byte-aligned quad mover with
    element-wise row ends), several segments long."""
    spacing, origin = (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)     # first shape: z lines of 150, y lines of 71
    f = (random_dvf(shape, spacing, seed=61, max_mm=5.0) + 0.5 * np.random.default_rng(62).normal(size=(3,) + shape)).astype(np.float32)
    sigma = [1.5, 1.5, 1.5]
    out = {}
    for two in ("", "1"):
        if two:
            monkeypatch.setenv("PP_RG_TWO_SWEEP", two)
        else:
            monkeypatch.delenv("PP_RG_TWO_SWEEP", raising=False)
        d = backend.dev(f)
        backend.ctx.recursive_gaussian_field(d, geom_of(shape, spacing, origin), sigma)
        out[two] = backend.host(d).copy()
    diff = np.abs(out[""] - out["1"])
    assert diff.max() <= 2e-6 and (diff > 0).mean() < 1e-3, (diff.max(), (diff > 0).mean())
    want = O.recursive_gaussian_vec(O.Vol(f.astype(np.float64), spacing, origin), sigma).arr
    np.testing.assert_allclose(out[""], want, rtol=0, atol=3e-6)


@pytest.mark.parametrize("shape", [(9, 21, 72), (20, 37, 136), (5, 16, 8), (33, 18, 64)])
def test_fused_discrete_gaussian_equals_the_three_passes(backend, shape, monkeypatch):
    """Round 4: DiscreteGaussian with radii <= 4 on rows of whole 16-byte strips runs as ONE kernel (k_gauss3_zyx: z pass in a
    register window, y pass through an LDS tile, x pass by DPP lane shifts) instead of three launches over the volume.  Same
    fmaf chains in the same order on fp32 intermediates: bit-identical to the separable passes (PP_GAUSS3=0), for equal and
    unequal radii per axis, tiles that overhang the volume, z-chunks, and volumes narrower than the rim."""
    spacing = (1.0, 1.3, 2.0)
    img = (phantom(shape, seed=11) + 300.0 * np.random.default_rng(5).standard_normal(shape)).astype(np.float32)
    size = (shape[2], shape[1], shape[0])
    for var in ((1.0, 1.0, 1.0), (4.0, 4.0, 4.0), (0.3, 2.5, 9.0), (2.25, 0.8, 30.0)):
        out = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("PP_GAUSS3", mode)
            o = backend.empty(shape)
            backend.ctx.discrete_gaussian(backend.dev(img), o, size, spacing, var, 0.01, 32, True)
            out[mode] = backend.host(o).copy()
        np.testing.assert_array_equal(out["0"], out["1"])
        assert np.abs(out["1"] - img).max() > 1.0


def test_fused_demons_pair_mix_visits_every_tile_once(backend, monkeypatch):
    """PP_PAIR_MIX: the second block of a CU (XCD-run index j >= 32) takes its x-neighbour's tile.  Only launches with more
    than 32 tiles per XCD ever swap, which the other tests' volumes never reach: 512 x 544 x 2 has 8 x 34 = 272 tiles of
    64 x 16 (34 per XCD), so ranks 32 / 33 of every XCD trade places.  A tile computed twice or never shows at once: against
    the ORACLE (and the staged schedule), MASK instances forced (pair priority and the progress words included)."""
    shape, spacing, origin = (2, 544, 512), (1.0, 1.0, 2.0), (0.0, 0.0, 0.0)
    fix = phantom(shape, seed=60)
    mov = (phantom(shape, seed=60, noise=0) + 30.0 * smooth_noise(shape, 61, cells=6)).astype(np.float32)
    p = _demons_params(backend.ctx, 2, spacing, _lib.DEMONS_FUSED, max_rms=0.0)
    monkeypatch.setenv("PP_FUSED_MASK", "1")
    monkeypatch.setenv("PP_FUSED_TILE", "0")
    f = backend.empty((3,) + shape)
    st = backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, f)
    got = backend.host(f).copy()
    flt = O.DemonsFilter()
    flt.SetSmoothUpdateField(True)
    flt.SetSmoothDisplacementField(True)
    flt.SetStandardDeviations([1.5 / s for s in spacing])
    flt.SetNumberOfIterations(2)
    flt.SetMaximumRMSError(0.0)
    want = flt.Execute(O.Vol(fix, spacing, origin), O.Vol(mov, spacing, origin)).arr
    err = np.abs(got - want)
    assert err.max() <= 2e-3 and np.sqrt((err ** 2).mean()) <= 5e-5, (err.max(), np.unravel_index(err.argmax(), err.shape))
    assert st.n_pixels == flt.stats.n_pixels == fix.size and st.elapsed_iterations == 2
    np.testing.assert_allclose(st.metric, flt.stats.metric, rtol=1e-6)
    monkeypatch.setenv("PP_FUSED_MASK", "0")      # the branchy instances on the same launch geometry: bit-identical
    f2 = backend.empty((3,) + shape)
    backend.ctx.demons_execute(backend.dev(fix), backend.dev(mov), geom_of(shape, spacing, origin), p, f2)
    np.testing.assert_array_equal(backend.host(f2).view(np.uint32), got.view(np.uint32))
