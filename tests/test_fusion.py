"""Drop-in label-fusion API (platipy_amd.label.fusion) against the oracle's restatement of
platipy/imaging/label/fusion.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.helpers import dice, phantom, smooth_noise

SHAPE, SPACING, ORIGIN = (16, 28, 44), (0.95, 1.05, 2.5), (3.0, -7.0, 11.0)


def _atlases(n=4):
    tgt = phantom(SHAPE, seed=200)
    out = []
    for i in range(n):
        ct = phantom(SHAPE, seed=200, noise=0) + (15 + 10 * i) * smooth_noise(SHAPE, 210 + i).astype(np.float32)
        lab = (smooth_noise(SHAPE, 220, cells=4) + 0.25 * smooth_noise(SHAPE, 230 + i, cells=5) > 0.1).astype(np.uint8)
        sub = (smooth_noise(SHAPE, 240, cells=5) + 0.2 * smooth_noise(SHAPE, 250 + i, cells=5) > 0.6).astype(np.uint8)
        out.append((ct.astype(np.float32), lab, sub))
    return tgt, out


@pytest.mark.parametrize("vote", ["unweighted", "global", "local", "block"])
def test_compute_weight_map(host_api, vote):
    pa = host_api
    tgt, atl = _atlases(1)
    want = O.compute_weight_map(O.Vol(tgt, SPACING, ORIGIN), O.Vol(atl[0][0], SPACING, ORIGIN), vote).arr
    got = pa.label.compute_weight_map(pa.image_from_array(tgt, SPACING, ORIGIN), pa.image_from_array(atl[0][0], SPACING, ORIGIN),
                                      vote_type=vote)
    assert got.tensor.dtype.is_floating_point and got.GetSize() == (44, 28, 16)
    # local: fp32 FIR of (T-M)^2 then 1/x; block: box mean then pow(-3); global: fp64 sum
    rtol = {"unweighted": 0, "global": 1e-6, "local": 3e-5, "block": 2e-4}[vote]
    np.testing.assert_allclose(got.numpy(), want, rtol=rtol, atol=0)
    with pytest.raises(NotImplementedError):
        pa.label.compute_weight_map(pa.image_from_array(tgt, SPACING, ORIGIN), pa.image_from_array(tgt, SPACING, ORIGIN),
                                    vote_type="patch_correlation")
    with pytest.raises(ValueError):
        pa.label.compute_weight_map(pa.image_from_array(tgt, SPACING, ORIGIN), pa.image_from_array(tgt, SPACING, ORIGIN),
                                    vote_type="nope")


def test_combine_labels_and_postprocess(host_api):
    pa = host_api
    tgt, atl = _atlases(4)
    aset_o, aset_g = {}, {}
    for i, (ct, lab, sub) in enumerate(atl):
        w = O.compute_weight_map(O.Vol(tgt, SPACING, ORIGIN), O.Vol(ct, SPACING, ORIGIN), "local")
        cid = f"{i:03d}"
        aset_o[cid] = {"DIR": {"Weight Map": w, "HEART": O.Vol(lab, SPACING, ORIGIN)}}
        aset_g[cid] = {"DIR": {"Weight Map": pa.image_from_array(w.arr, SPACING, ORIGIN),
                               "HEART": pa.image_from_array(lab, SPACING, ORIGIN)}}
        if i != 2:  # one atlas lacks the sub-structure (fusion.py:251-252)
            aset_o[cid]["DIR"]["SUB"] = O.Vol(sub, SPACING, ORIGIN)
            aset_g[cid]["DIR"]["SUB"] = pa.image_from_array(sub, SPACING, ORIGIN)
    want = O.combine_labels(aset_o, ["HEART", "SUB"])
    got = pa.label.combine_labels(aset_g, ["HEART", "SUB"])
    assert set(got) == {"HEART", "SUB"}
    for k in want:
        g = got[k].numpy()
        # fp32 weighted sums (w * L may be contracted to fma on the GPU), divide, 1-voxel blur, rescale
        np.testing.assert_allclose(g, want[k].arr, rtol=0, atol=5e-6)
        assert g.max() == 1.0 and g.min() == 0.0
        assert ((g > 0) & (g < 1e-4)).sum() == 0      # Threshold(lower=1e-4) zeroed the tail
        wm = O.process_probability_image(want[k], 0.5).arr
        gm = pa.label.process_probability_image(got[k], 0.5).numpy()
        assert gm.dtype == np.uint8
        np.testing.assert_array_equal(gm, wm)          # binary result: bit-exact
        assert 0 < gm.sum() < gm.size
    # single structure given as str
    one = pa.label.combine_labels(aset_g, "HEART")
    np.testing.assert_array_equal(one["HEART"].numpy(), got["HEART"].numpy())
    assert dice(pa.label.process_probability_image(one["HEART"]).numpy(), atl[0][1]) > 0.8


def test_process_probability_image_edge_cases(host_api):
    pa = host_api
    z = np.zeros(SHAPE, np.float32)
    z[2:5, 3:9, 4:12] = 0.9   # large blob with a hole
    z[3, 5:7, 6:9] = 0.0
    z[10:12, 20:22, 30:33] = 1.0  # small, brighter blob
    want = O.process_probability_image(O.Vol(z, SPACING, ORIGIN), 0.5).arr
    got = pa.label.process_probability_image(pa.image_from_array(z, SPACING, ORIGIN), 0.5).numpy()
    np.testing.assert_array_equal(got, want)
    assert got[3, 5, 7] == 1 and got[10, 20, 30] == 0  # hole filled, small component dropped
    # nothing above threshold -> the (empty) binary image comes back
    e = np.full(SHAPE, 0.1, np.float32)
    e[0, 0, 0] = 1.0
    got = pa.label.process_probability_image(pa.image_from_array(e, SPACING, ORIGIN), 0.5).numpy()
    assert got.sum() == 1


def test_overlap_correction_and_distance_map_helpers(host_api):
    pa = host_api
    a = np.zeros(SHAPE, np.uint8)
    b = np.zeros(SHAPE, np.uint8)
    c = np.zeros(SHAPE, np.uint8)
    a[2:12, 4:20, 6:30] = 1          # largest
    b[8:14, 10:24, 20:40] = 1        # overlaps a
    c[9:11, 12:16, 24:28] = 1        # inside both
    imgs = {k: pa.image_from_array(v, SPACING, ORIGIN) for k, v in (("B", b), ("A", a), ("C", c))}
    out = pa.label.correct_volume_overlap(imgs)
    oa, ob, oc = out["A"].numpy(), out["B"].numpy(), out["C"].numpy()
    np.testing.assert_array_equal(oa, a)                          # the largest keeps everything
    np.testing.assert_array_equal(ob, b & ~a)
    np.testing.assert_array_equal(oc, c & ~a & ~b)
    assert (oa + ob + oc).max() == 1
    small_first = pa.label.correct_volume_overlap(imgs, assign_overlap_to_largest=False)
    np.testing.assert_array_equal(small_first["C"].numpy(), c)
    # inside-positive signed distance map and the registration structure built from it
    dm = pa.registration.convert_mask_to_distance_map(imgs["A"]).numpy()
    want = O.maurer_distance_map(O.Vol(a, SPACING, ORIGIN), signed=True, inside_positive=True).arr
    np.testing.assert_allclose(dm, want, rtol=2e-6, atol=2e-5)
    assert dm[6, 10, 15] > 0 > dm[0, 0, 0]
    rs = pa.registration.convert_mask_to_reg_structure(imgs["A"]).numpy()
    assert rs.max() == 1.0 and rs[0, 0, 0] == 0.0
    # expansion in mm -> ball dilation by int(mm / spacing) voxels per axis before the distance map (utils.py:326-329)
    rs2 = pa.registration.convert_mask_to_reg_structure(imgs["A"], expansion=2)
    radius = [int(2 / sp) for sp in SPACING]
    grown = O.binary_dilate_ball(O.Vol(a.astype(np.uint8), SPACING, ORIGIN), radius)
    want2 = O.maurer_distance_map(grown, signed=True, inside_positive=True).arr.astype(np.float64) * (grown.arr != 0)
    assert rs2.tensor.dtype == torch.float64
    np.testing.assert_allclose(rs2.numpy(), want2 / want2.max(), rtol=1e-5, atol=1e-6)
    # the pipelines' inline mask calls: closing after "largest component" (multiatlas/run.py:421-423)
    lc = pa.label.utils.largest_component(imgs["C"]).numpy()
    assert lc.max() == 1 and (lc <= c).all()
    closed = pa.label.utils.binary_morphological_closing(imgs["A"], [1, 1, 1]).numpy()
    np.testing.assert_array_equal(closed, O.binary_closing_ball(O.Vol(a.astype(np.uint8), SPACING, ORIGIN), [1, 1, 1]).arr)


def test_combine_labels_weights_float_and_wide_integer_labels_as_float32(host_api):
    """The reference casts every label to sitkFloat32 before weighting (label/fusion.py:269-272): a probabilistic label of
    0.7 contributes 0.7 w and an int16 label of 256 contributes 256 w -- neither is squeezed through uint8."""
    pa = host_api
    rng = np.random.default_rng(3)
    shape, sp = (6, 9, 10), (1.0, 1.0, 1.0)
    w = [rng.uniform(0.5, 2.0, shape).astype(np.float32) for _ in range(2)]
    soft = [rng.uniform(0.0, 1.0, shape).astype(np.float32) for _ in range(2)]
    aset = {f"{k}": {"DIR": {"Weight Map": pa.image_from_array(w[k], sp), "S": pa.image_from_array(soft[k], sp)}} for k in range(2)}
    got = pa.label.combine_labels(aset, "S", threshold=0.0, smooth_sigma=1e-3)["S"].numpy()
    want = (w[0] * soft[0] + w[1] * soft[1]) / (w[0] + w[1])
    want = (want - want.min()) / (want.max() - want.min())          # RescaleIntensity(0, 1)
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-6)
    wide = [np.where(rng.uniform(size=shape) > 0.5, 256, 0).astype(np.int16) for _ in range(2)]
    aset = {f"{k}": {"DIR": {"Weight Map": pa.image_from_array(w[k], sp), "S": pa.image_from_array(wide[k], sp)}} for k in range(2)}
    got = pa.label.combine_labels(aset, "S", threshold=0.0, smooth_sigma=1e-3)["S"].numpy()
    assert got.max() == 1.0 and (got > 0).mean() > 0.5            # 256 did not wrap to 0


@pytest.mark.gpu
def test_process_probability_image_on_the_support_box_equals_the_whole_volume(monkeypatch):
    """Round 6: on volumes of >= 4 Mvoxel process_probability_image works on the box around the probability's support (a fused
    probability is zero outside the smoothed union of the atlas labels).  The result is that of the whole volume, bit for bit:
    a blob with an internal hole, a second smaller component, one touching the volume's border."""
    import torch

    import platipy_amd as pa
    from platipy_amd.label import fusion

    n = 168
    zz, yy, xx = np.meshgrid(*[np.arange(n, dtype=np.float32)] * 3, indexing="ij")
    prob = np.zeros((n, n, n), np.float32)
    r = np.sqrt((xx - 70) ** 2 + (yy - 80) ** 2 + (zz - 60) ** 2)
    prob[r < 30] = 0.9
    prob[r < 8] = 0.0                                      # a hole inside the blob
    prob[(np.abs(xx - 120) < 6) & (np.abs(yy - 30) < 6) & (np.abs(zz - 100) < 6)] = 0.7      # a smaller component
    prob[0:5, 100:110, 100:110] = 0.6                      # ... and one on the border
    prob += np.where(prob > 0, 0.05 * np.sin(xx * 0.3), 0).astype(np.float32)
    img = pa.image_from_array(prob, (1.0, 1.1, 1.2))
    got = pa.label.process_probability_image(img, 0.5).numpy()
    monkeypatch.setattr(fusion, "CROP_MIN_VOXELS", 1 << 62)
    whole = pa.label.process_probability_image(img, 0.5).numpy()
    want = O.process_probability_image(O.Vol(prob, (1.0, 1.1, 1.2)), 0.5).arr
    assert got.sum() > 50000 and got[60, 80, 70] == 1            # the hole is filled, the largest component kept
    np.testing.assert_array_equal(got, whole)
    np.testing.assert_array_equal(got, want)
    assert torch.cuda.is_available()
