#!/usr/bin/env python
"""Regenerate tests/golden/*.npz.

BUILD-DERIVED, NOT REFERENCE-DERIVED: the reference's arithmetic (SimpleITK) cannot run in the build image and its
tests hold no vectors for this path, so these fixtures freeze the *oracle's* outputs (oracle/, the CPU restatement of
the ITK filters) on small seeded inputs.  They pin both the oracle and the product against silent drift; they say
nothing new about parity with SimpleITK ("parity unpinned").  Inputs are generated here from numpy seeds
only -- no file of the reference is read."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import linear_oracle, oracle as O  # noqa: E402

SHAPE, SPACING, ORIGIN = (10, 14, 18), (0.9, 1.1, 2.5), (320.0, -52.0, 60.0)


def inputs():
    rng = np.random.default_rng(20260927)
    zz, yy, xx = np.meshgrid(*[np.arange(n) for n in SHAPE], indexing="ij")
    fixed = (-1000 + 1000 * np.exp(-(((xx - 9) / 5.0) ** 2 + ((yy - 7) / 4.0) ** 2 + ((zz - 5) / 3.0) ** 2)) + rng.normal(0, 3, SHAPE)).astype(np.float32)
    moving = (-1000 + 1000 * np.exp(-(((xx - 10.2) / 5.0) ** 2 + ((yy - 6.4) / 4.2) ** 2 + ((zz - 5.3) / 3.0) ** 2)) + rng.normal(0, 3, SHAPE)).astype(np.float32)
    field = (rng.normal(0, 1, (3, 3, 4, 5))).astype(np.float64)
    from scipy.ndimage import zoom

    field = np.stack([zoom(field[c], [SHAPE[i] / field.shape[i + 1] for i in range(3)], order=1)[: SHAPE[0], : SHAPE[1], : SHAPE[2]] for c in range(3)])
    field = (field * 1.5).astype(np.float32)
    mask = ((xx - 9) ** 2 / 30.0 + (yy - 7) ** 2 / 20.0 + (zz - 5) ** 2 / 8.0 < 1).astype(np.uint8)
    return fixed, moving, field, mask


def NOTCHED(mask):
    """the mask with a slot cut into it, so that closing has something to close"""
    m = mask.copy()
    m[4:6, 6:8, 3:16] = 0
    return m


# (Af, bf, Am, bm, vsize, stride): virtual 9x7x5 lattice over the 18x14x10 images, a slightly sheared moving map
METRIC_MAP = (np.eye(3) * 2.0, np.array([0.5, 0.5, 0.5]),
              np.array([[1.9137, 0.1071, 0.0031], [-0.0813, 2.0519, 0.0207], [0.0109, 0.0043, 1.9011]]), np.array([0.7123, -0.4057, 0.4131]),
              (9, 7, 5), 2)


def main():
    fixed, moving, field, mask = inputs()
    vf, vm = O.Vol(fixed, SPACING, ORIGIN), O.Vol(moving, SPACING, ORIGIN)
    f64 = field.astype(np.float64)
    warped = O.warp_image(vm, f64).arr
    upd, st = O.esm_update(vf, O.Vol(warped, SPACING, ORIGIN))
    flt = O.DemonsFilter()
    flt.SetSmoothUpdateField(True)
    flt.SetStandardDeviations([1.5 / s for s in SPACING])
    flt.SetNumberOfIterations(3)
    flt.SetMaximumRMSError(0.0)
    execd = flt.Execute(vf, vm).arr
    out = {
        "fixed": fixed, "moving": moving, "field": field, "mask": mask,
        "taps_var1_err0p1": O.gaussian_operator(1.0, 0.1, 30),
        "taps_var2p25_err0p1": O.gaussian_operator(2.25, 0.1, 30),
        "taps_var4_err0p01": O.gaussian_operator(4.0, 0.01, 32),
        "discrete_gaussian_var4": O.discrete_gaussian(vf, 4.0).arr,
        "smooth_field": O.smooth_field(f64, [1.5 / s for s in SPACING]).astype(np.float32),
        "warp_sentinel": warped,
        "esm_update": upd.astype(np.float32),
        "esm_stats": np.array([st.metric, st.rms_change, st.n_pixels], dtype=np.float64),
        "execute_3it": execd.astype(np.float32),
        "execute_stats": np.array([flt.GetMetric(), flt.GetRMSChange(), flt.GetElapsedIterations()], dtype=np.float64),
        "recursive_gaussian": O.recursive_gaussian_vec(O.Vol(f64, SPACING, ORIGIN), [1.5 / s for s in SPACING]).arr.astype(np.float32),
        "mask_nn_through_field": O.resample(O.Vol(mask, SPACING, ORIGIN), O.Vol(mask, SPACING, ORIGIN), field_vol=O.Vol(f64, SPACING, ORIGIN),
                                            interp=O.INTERP_NEAREST).arr,
        "weight_local": O.compute_weight_map(vf, vm, "local").arr,
        "distance_map_signed": O.maurer_distance_map(O.Vol(mask, SPACING, ORIGIN), signed=True).arr,
        "label_contour": O.label_contour(O.Vol(mask, SPACING, ORIGIN)).arr,
        "dilate_ball_221": O.binary_dilate_ball(O.Vol(mask, SPACING, ORIGIN), (2, 2, 1)).arr,
        "erode_ball_111": O.binary_erode_ball(O.Vol(mask, SPACING, ORIGIN), (1, 1, 1)).arr,
        "close_ball_210": O.binary_closing_ball(O.Vol(NOTCHED(mask), SPACING, ORIGIN), (2, 1, 0)).arr,
        "meansq_affine": np.array(linear_oracle.meansq_affine(fixed, moving, *METRIC_MAP), dtype=np.float64),
    }
    np.savez_compressed(os.path.join(HERE, "hotpath_small.npz"), **out)
    print("wrote", os.path.join(HERE, "hotpath_small.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
