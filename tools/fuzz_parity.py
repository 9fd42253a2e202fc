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
print("cases out of tolerance:", bad)
sys.exit(1 if bad else 0)
