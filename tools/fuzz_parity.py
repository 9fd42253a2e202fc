#!/usr/bin/env python
"""Randomised parity sweep on a GPU box: demons Execute (fused and staged), resampling through a field and the fusion
arithmetic against the CPU oracle over random sizes / spacings / schedules.  Prints one line per case and a summary;
exits non-zero if any case leaves the tolerances the test-suite states."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from platipy_amd import _lib  # noqa: E402
from tests.helpers import phantom, random_dvf  # noqa: E402

ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for case in range(n_cases):
    shape = tuple(int(v) for v in rng.integers(9, 97, size=3))
    spacing = tuple(float(v) for v in rng.uniform(0.6, 2.6, size=3))
    origin = tuple(float(v) for v in rng.uniform(-100, 100, size=3))
    iters = int(rng.integers(1, 9))
    sigma_mm = float(rng.uniform(0.8, 2.5))
    fix = phantom(shape, seed=1000 + case)
    dv = random_dvf(shape, spacing, seed=2000 + case, max_mm=float(rng.uniform(0.5, 4.0)))
    mov = O.warp_image(O.Vol(phantom(shape, seed=1000 + case, noise=0), spacing, origin), dv.astype(np.float64), edge_value=-1000.0).arr
    mov = (mov + rng.normal(0, 5, size=shape)).astype(np.float32)
    flt = O.DemonsFilter()
    flt.SetSmoothUpdateField(True)
    flt.SetStandardDeviations([sigma_mm / s for s in spacing])
    flt.SetNumberOfIterations(iters)
    want = flt.Execute(O.Vol(fix, spacing, origin), O.Vol(mov, spacing, origin)).arr
    g = _lib.make_geom(shape[::-1], spacing, origin)
    line = f"case {case:2d} shape {shape} spacing {tuple(round(s, 2) for s in spacing)} it {iters} sigma {sigma_mm:.2f} mm:"
    for variant, name in ((_lib.DEMONS_FUSED, "fused"), (_lib.DEMONS_STAGED, "staged")):
        p = ctx.default_demons_params()
        p.smooth_update, p.iterations, p.variant = 1, iters, variant
        p.sigma_d_vox[:] = [sigma_mm / s for s in spacing]
        field = torch.zeros((3,) + shape, device="cuda")
        try:
            st = ctx.demons_execute(torch.from_numpy(fix).cuda(), torch.from_numpy(mov).cuda(), g, p, field)
        except _lib.PlatipyAmdError as e:
            line += f"  {name}: {e}"
            continue
        err = np.abs(field.cpu().numpy() - want)
        ok = st.elapsed_iterations == flt.GetElapsedIterations() and err.max() <= 5e-3 and np.sqrt((err ** 2).mean()) <= 1e-4
        bad += 0 if ok else 1
        line += f"  {name}: max {err.max():.2e} rms {np.sqrt((err ** 2).mean()):.2e} it {st.elapsed_iterations}/{flt.GetElapsedIterations()}{'' if ok else '  <-- OUT OF TOLERANCE'}"
    # NN label propagation through the same field: bit-exact
    mask = (phantom(shape, seed=3000 + case, noise=0) > -300).astype(np.uint8)
    wantm = O.resample(O.Vol(mask, spacing, origin), O.Vol(mask, spacing, origin), field_vol=O.Vol(dv.astype(np.float64), spacing, origin),
                       interp=O.INTERP_NEAREST).arr
    out = torch.zeros(shape, dtype=torch.uint8, device="cuda")
    ctx.resample(torch.from_numpy(mask).cuda(), g, g, out, field=torch.from_numpy(dv).cuda(), interp=_lib.INTERP_NEAREST, default_value=0.0,
                 u8=True)
    same = bool((out.cpu().numpy() == wantm).all())
    bad += 0 if same else 1
    print(line + f"  mask NN bit-exact: {same}")

# ---- second sweep: the once-per-level kernels (pyramid blur + resample, field resample / compose / recursive Gaussian,
# local weight map) and the label kernels (distance map, contour, fill-hole + largest component, ball morphology)
import platipy_amd as pa  # noqa: E402

for case in range(max(4, n_cases // 2)):
    shape = tuple(int(v) for v in rng.integers(12, 70, size=3))
    spacing = tuple(float(v) for v in rng.uniform(0.7, 2.6, size=3))
    origin = tuple(float(v) for v in rng.uniform(-100, 100, size=3))
    img = phantom(shape, seed=5000 + case)
    vol = O.Vol(img, spacing, origin)
    pimg = pa.image_from_array(img, spacing, origin)
    line = f"case {case:2d} shape {shape}:"
    shrink = int(rng.integers(2, 5))
    sig = float(rng.uniform(1.0, 4.0))
    want = O.smooth_and_resample(vol, None, shrink, sig).arr
    got = pa.registration.smooth_and_resample(pimg, shrink_factor=shrink, smoothing_sigma=sig).numpy()
    e1 = float(np.abs(got - want).max())
    dv = random_dvf(shape, spacing, seed=6000 + case, max_mm=3.0)
    sig_mm = [1.5 / s for s in spacing]
    want = O.recursive_gaussian_vec(O.Vol(dv.astype(np.float64), spacing, origin), sig_mm).arr
    f = torch.from_numpy(dv).cuda()
    ctx.recursive_gaussian_field(f, _lib.make_geom(shape[::-1], spacing, origin), sig_mm)      # in place
    e2 = float(np.abs(f.cpu().numpy() - want).max())
    mov = (img + rng.normal(0, 20, size=shape)).astype(np.float32)
    want = O.compute_weight_map(vol, O.Vol(mov, spacing, origin), "local").arr
    got = pa.label.compute_weight_map(pimg, pa.image_from_array(mov, spacing, origin), vote_type="local").numpy()
    e3 = float(np.abs(got / want - 1).max())
    mask = (phantom(shape, seed=7000 + case, noise=0) > -250).astype(np.uint8)
    mask[:, :2, :] = 0
    mv, pm = O.Vol(mask, spacing, origin), pa.image_from_array(mask, spacing, origin)
    e4 = float(np.abs(pa.label.distance_map(pm, signed=True).numpy() - O.maurer_distance_map(mv, signed=True).arr).max())
    same_c = bool((pa.label.label_contour(pm).numpy() == O.label_contour(mv).arr).all())
    radius = [int(v) for v in rng.integers(0, 4, size=3)]
    same_m = bool((pa.label.binary_morphological_closing(pm, radius).numpy() == O.binary_closing_ball(mv, radius).arr).all()) and \
        bool((pa.label.binary_dilate(pm, radius).numpy() == O.binary_dilate_ball(mv, radius).arr).all())
    prob = O.discrete_gaussian(O.Vol(mask.astype(np.float32), spacing, origin), 2.0).arr
    same_p = bool((pa.label.process_probability_image(pa.image_from_array(prob, spacing, origin), 0.5).numpy() ==
                   O.process_probability_image(O.Vol(prob, spacing, origin), 0.5).arr).all())
    ok = e1 <= 3e-3 and e2 <= 2e-4 and e3 <= 2e-4 and e4 <= 2e-4 and same_c and same_m and same_p
    bad += 0 if ok else 1
    print(line + f" pyramid level max {e1:.1e}  recursive Gaussian max {e2:.1e}  weight map rel {e3:.1e}  distance map max {e4:.1e}"
          f"  contour {same_c}  morphology {same_m}  probability->mask {same_p}{'' if ok else '  <-- OUT OF TOLERANCE'}")
# ---- third sweep (round 6): WHOLE registrations with the pipelines' settings -- isotropic levels of random voxel size, sigma 0,
# many iterations with the RMS halt live, final resample onto the (anisotropic) fixed grid -- and the structure-guided stage on
# distance-map images (multiatlas/run.py:75-84, cardiac/run.py:129-152); tolerance = tests/test_pipeline_parity.py's
from tests.test_pipeline_parity import ellipsoid, err_stats  # noqa: E402

for case in range(max(3, n_cases // 3)):
    shape = tuple(int(v) for v in rng.integers(40, 90, size=3))
    spacing = (float(rng.uniform(0.8, 1.3)),) * 2 + (float(rng.uniform(1.5, 3.0)),)
    origin = tuple(float(v) for v in rng.uniform(-100, 100, size=3))
    fine = float(rng.uniform(1.4, 2.2))
    guided = case % 3 == 2
    kw = dict(isotropic_resample=True, resolution_staging=[4 * fine, 2 * fine, fine], smoothing_sigmas=[0, 0, 0],
              iteration_staging=[int(v) for v in rng.integers(30, 120, size=3)], default_value=0 if guided else None)
    if guided:
        c = [s / 2 for s in shape[::-1]]
        t = ellipsoid(shape, c, [0.3 * s for s in shape[::-1]])
        a = ellipsoid(shape, [v + float(rng.uniform(-4, 4)) for v in c], [float(rng.uniform(0.22, 0.34)) * s for s in shape[::-1]])
        fix = pa.registration.convert_mask_to_reg_structure(pa.Image(t, spacing, origin), expansion=2).numpy()
        mov = pa.registration.convert_mask_to_reg_structure(pa.Image(a, spacing, origin), expansion=2).numpy()
    else:
        fix = phantom(shape, seed=8000 + case)
        dv = random_dvf(shape, spacing, seed=8100 + case, max_mm=float(rng.uniform(2.0, 6.0)))
        mov = O.warp_image(O.Vol(phantom(shape, seed=8000 + case, noise=0), spacing, origin), dv.astype(np.float64), edge_value=-1000.0).arr
        mov = (mov + rng.normal(0, 5, size=shape)).astype(np.float32)
    tr, ptr = [], []
    _, w, _ = O.fast_symmetric_forces_demons_registration(O.Vol(fix, spacing, origin), O.Vol(mov, spacing, origin), trace=tr, **kw)
    # the oracle's own conditioning: its response to the moving image one ulp up and one ulp down; the larger of the two per
    # statistic (a maximum over one trial of a chaotic iteration is a noisy estimate: 0.1 ... 1.0 mm across this sweep's cases)
    pert = np.nextafter(mov.astype(np.float32), np.float32(np.inf))
    _, p, _ = O.fast_symmetric_forces_demons_registration(O.Vol(fix, spacing, origin), O.Vol(pert, spacing, origin), trace=ptr, **kw)
    pert = np.nextafter(mov.astype(np.float32), np.float32(-np.inf))
    _, p2, _ = O.fast_symmetric_forces_demons_registration(O.Vol(fix, spacing, origin), O.Vol(pert, spacing, origin), **kw)
    _, tfm, g = pa.registration.fast_symmetric_forces_demons_registration(pa.image_from_array(fix, spacing, origin),
                                                                          pa.image_from_array(mov, spacing, origin), **kw)
    hip, own, own2 = err_stats(g.numpy(), w.arr), err_stats(p.arr, w.arr), err_stats(p2.arr, w.arr)
    own = {k: max(own[k], own2[k]) for k in own}
    mask = ellipsoid(shape, [s / 2 for s in shape[::-1]], [0.25 * s for s in shape[::-1]]).cpu().numpy()
    mh = pa.registration.apply_transform(pa.image_from_array(mask, spacing, origin), transform=tfm, default_value=0,
                                         interpolator=pa.sitkNearestNeighbor).numpy()
    mo = O.apply_transform(O.Vol(mask, spacing, origin), field_vol=w, default_value=0, interpolator=O.INTERP_NEAREST).arr
    mp = O.apply_transform(O.Vol(mask, spacing, origin), field_vol=p, default_value=0, interpolator=O.INTERP_NEAREST).arr
    mp2 = O.apply_transform(O.Vol(mask, spacing, origin), field_vol=p2, default_value=0, interpolator=O.INTERP_NEAREST).arr
    ndiff, nown = int((mh != mo).sum()), max(int((mp != mo).sum()), int((mp2 != mo).sum()))
    ok = (hip["median"] <= max(5e-5, 4 * own["median"]) and hip["p99"] <= max(1e-3, 4 * own["p99"]) and hip["rms"] <= max(2e-3, 4 * own["rms"])
          and hip["inner_max"] <= max(2e-2, 4 * own["inner_max"]) and ndiff <= max(2e-4 * mask.sum(), 4 * nown, 2))
    bad += 0 if ok else 1
    print(f"case {case:2d} {'guided' if guided else 'ct    '} shape {shape} levels {[t['fixed'].arr.shape[::-1] for t in tr]} iterations "
          f"{[t['elapsed'] for t in tr]}: field median {hip['median']:.1e} (own {own['median']:.1e}) p99 {hip['p99']:.1e} ({own['p99']:.1e}) "
          f"rms {hip['rms']:.1e} ({own['rms']:.1e}) inner max {hip['inner_max']:.1e} ({own['inner_max']:.1e}); whole-chain mask voxels differing "
          f"{ndiff} (own {nown}) of {int(mask.sum())}"
          f"{'' if ok else '  <-- OUT OF TOLERANCE'}")
print("cases out of tolerance:", bad)
sys.exit(1 if bad else 0)
