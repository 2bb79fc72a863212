"""CPU tests: the oracle against SURVEY.md 8c known-answer values, the committed golden
fixtures, the independent plain-C restatement, and property checks (reference has no tests:
SURVEY.md section 4)."""
import os

import numpy as np
import pytest

import helpers
from oracle import bbox_oracle as bo
from oracle import c_oracle as co

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_scales_a1():
    s = [bo.get_scale_for_nth_feature_map(k) for k in range(1, 8)]
    np.testing.assert_allclose(s, [0.2, 0.34, 0.48, 0.62, 0.76, 0.9, 1.04], rtol=0, atol=1e-12)


@pytest.mark.parametrize("backbone,n,total", [("mobilenet_v2", 2268, 4535.999987), ("vgg16", 8732, 17463.999998)])
def test_priors_known_answers(backbone, n, total):
    """SURVEY.md 8c known-answer values (shape, f64 checksum, first/last rows)."""
    p = bo.generate_prior_boxes(helpers.FMAPS[backbone], helpers.ASPECT_RATIOS)
    assert p.shape == (n, 4) and p.dtype == np.float32
    assert abs(p.astype(np.float64).sum() - total) < 5e-6
    last4 = np.array([[0.050000012, 0.050000012, 0.95, 0.95], [0.18180194, 0, 0.8181981, 1],
                      [0, 0.18180197, 1, 0.818198], [0.016264528, 0.016264528, 0.98373544, 0.98373544]], np.float32)
    np.testing.assert_array_equal(p[-4:], last4)
    if backbone == "mobilenet_v2":
        first4 = np.array([[0, 0, 0.12631579, 0.12631579], [0, 0, 0.09702647, 0.16773716],
                           [0, 0, 0.16773716, 0.09702647], [0, 0, 0.15669985, 0.15669985]], np.float32)
    else:
        first4 = np.array([[0, 0, 0.1131579, 0.1131579], [0, 0, 0.08386858, 0.15457925],
                           [0, 0, 0.15457925, 0.08386858], [0, 0, 0.14354195, 0.14354195]], np.float32)
    np.testing.assert_array_equal(p[:4], first4)
    gold = np.load(os.path.join(GOLD, "priors.npz"))[backbone]
    np.testing.assert_array_equal(p, gold)


def test_prior_level_offsets():
    offs = np.cumsum([0] + [f * f * (len(a) + 1) for f, a in zip(helpers.FMAPS["mobilenet_v2"], helpers.ASPECT_RATIOS)])
    assert list(offs) == [0, 1444, 2044, 2194, 2248, 2264, 2268]
    offs = np.cumsum([0] + [f * f * (len(a) + 1) for f, a in zip(helpers.FMAPS["vgg16"], helpers.ASPECT_RATIOS)])
    assert list(offs) == [0, 5776, 7942, 8542, 8692, 8728, 8732]


def test_decode_encode_roundtrip():
    p = bo.generate_prior_boxes(helpers.FMAPS["mobilenet_v2"], helpers.ASPECT_RATIOS)
    d, _ = helpers.decoder_inputs(2, p.shape[0], seed=11)
    d *= 0.3
    boxes = bo.get_bboxes_from_deltas(p, d)
    back = bo.get_deltas_from_bboxes(p, boxes)
    np.testing.assert_allclose(back, d, atol=2e-4)
    np.testing.assert_allclose(co.decode(p, d, [1, 1, 1, 1]), boxes, atol=3e-7)


def test_golden_decode_nms_numpy_and_c():
    z = np.load(os.path.join(GOLD, "decode_nms.npz"))
    p = np.load(os.path.join(GOLD, "priors.npz"))["mobilenet_v2"]
    for name in ("rand", "none", "ties", "degenerate"):
        d, pr = z[name + "_deltas"], z[name + "_probs"]
        b, l, s, v, i = bo.ssd_decode(p, helpers.VARIANCES, d, pr, return_indices=True)
        np.testing.assert_array_equal(i, z[name + "_idx"])
        np.testing.assert_array_equal(v, z[name + "_valid"])
        np.testing.assert_array_equal(b, z[name + "_boxes"])
        cb, cl, cs, cv, ci = co.decode_nms(d, pr, p, helpers.VARIANCES)
        np.testing.assert_array_equal(ci, i)
        np.testing.assert_array_equal(cv, v)
        np.testing.assert_array_equal(cl, l)
        np.testing.assert_array_equal(cs, s)
        np.testing.assert_allclose(cb, b, atol=3e-7)
    assert z["none_valid"][0] == 0 and not z["none_boxes"].any()
    assert z["rand_valid"].max() == 200


def test_nms_properties():
    """sorted by score, rows >= valid are zero, no kept same-class pair with IoU > 0.5,
    boxes clipped, labels never 0 for softmax inputs."""
    p = bo.generate_prior_boxes(helpers.FMAPS["mobilenet_v2"], helpers.ASPECT_RATIOS)
    d, pr = helpers.decoder_inputs(3, p.shape[0], seed=21, boost_frac=0.05)
    b, l, s, v, i = co.decode_nms(d, pr, p, helpers.VARIANCES)
    dec = co.decode(p, d, helpers.VARIANCES)
    for k in range(3):
        n = v[k]
        assert (np.diff(s[k, :n]) <= 0).all()
        assert not b[k, n:].any() and not s[k, n:].any() and not l[k, n:].any()
        assert (i[k, n:] == -1).all()
        assert b.min() >= 0 and b.max() <= 1
        assert (l[k, :n] >= 1).all() and (s[k, :n] > 0.5).all()
        for c in np.unique(l[k, :n]):
            idx = i[k, :n][l[k, :n] == c]
            for a in range(len(idx)):
                for bb in range(a + 1, len(idx)):
                    assert bo._nms_iou(dec[k, idx[a]], dec[k, idx[bb]]) <= 0.5


def test_combined_nms_column0_and_threshold_semantics():
    """Column 0 is an ordinary class for NMS; thresholds are strict (Appendix B.1/B.2)."""
    boxes = np.array([[[0, 0, 1, 1], [0, 0, 1, 0.5], [0.5, 0.5, 1, 1], [0, 0, 0.5, 0.5]]], np.float32)
    scores = np.zeros((1, 4, 3), np.float32)
    scores[0, :, 0] = [0.9, 0.8, 0.5, 0.7]   # 0.5 is NOT > 0.5
    scores[0, :, 2] = [0.6, 0.0, 0.0, 0.0]
    b, s, c, v, i = bo.combined_non_max_suppression(boxes[:, :, None], scores, 200, 200, 0.5, 0.5,
                                                    return_indices=True)
    # class 0: box0 (0.9) kept; box1 IoU with box0 = 0.5 -> NOT suppressed (needs > 0.5); box3 IoU .25 kept
    assert v[0] == 4
    np.testing.assert_array_equal(i[0, :4], [0, 1, 3, 0])
    np.testing.assert_array_equal(c[0, :4], [0, 0, 0, 2])


def test_golden_match_numpy_and_c():
    z = np.load(os.path.join(GOLD, "match.npz"))
    p = np.load(os.path.join(GOLD, "priors.npz"))["mobilenet_v2"]
    hp = helpers.hyper_params()
    dl, oh, lab, mi = bo.calculate_actual_outputs(p, z["gt"], z["gl"], hp, return_indices=True)
    np.testing.assert_array_equal(lab, z["label_idx"])
    np.testing.assert_array_equal(mi, z["match_idx"])
    np.testing.assert_array_equal(dl, z["deltas"])
    assert oh.shape == (4, 2268, 21) and (oh.sum(-1) == 1).all()
    assert (oh.argmax(-1) == lab).all()
    cd, cl, cm = co.match_encode(p, z["gt"], z["gl"], helpers.VARIANCES)
    np.testing.assert_array_equal(cl, lab)
    np.testing.assert_array_equal(cm, mi)
    np.testing.assert_allclose(cd, dl, atol=2e-6)
    np.testing.assert_array_equal(co.iou_map(p, z["gt"]), z["iou"])
    # positives exist, padded labels (-1) never selected
    assert (lab > 0).sum() > 0 and (lab >= 0).all()
    # deltas are zero exactly where not positive (ssd_loss.py:25 relies on it)
    assert not dl[lab == 0].any()


def test_iou_degenerate_nan_like_reference():
    """No epsilon: two zero-area boxes give 0/0 = NaN (utils/bbox_utils.py:59)."""
    z = np.zeros((1, 1, 4), np.float32)
    out = bo.generate_iou_map(np.zeros((1, 4), np.float32), z)
    assert np.isnan(out).all()
    assert np.isnan(co.iou_map(np.zeros((1, 4), np.float32), z)).all()


def test_normalize_helpers():
    b = np.array([[[10.5, 20.5, 30.5, 41.5]]], np.float32)
    n = bo.normalize_bboxes(b, 100, 200)
    np.testing.assert_allclose(n, [[[0.105, 0.1025, 0.305, 0.2075]]], rtol=1e-6)
    d = bo.denormalize_bboxes(np.array([[[0.105, 0.1025, 0.305, 0.2075]]], np.float32), 100, 200)
    np.testing.assert_array_equal(d, [[[10, 20, 30, 42]]])   # 10.5 -> 10, 20.5 -> 20 (half to even)
    r = bo.renormalize_bboxes_with_min_max(np.array([[0.2, 0.2, 0.6, 1.2]], np.float32),
                                           np.array([0.1, 0.1, 0.9, 0.9], np.float32))
    np.testing.assert_allclose(r, [[0.125, 0.125, 0.625, 1.0]], rtol=1e-6)


def test_tf_published_nms_known_answers():
    """External anchors for the CombinedNMS restatement: the expected selections of TensorFlow's
    own NMS unit tests ([3P], from memory -- see helpers.tf_nms_known_answers)."""
    import helpers
    for c in helpers.tf_nms_known_answers():
        N = c["boxes"].shape[0]
        b, s, k, v, idx = bo.combined_non_max_suppression(c["boxes"][None, :, None, :], c["scores"][None], c["mpc"],
                                                          c["mt"], c["iou"], c["thr"], clip_boxes=c["clip"],
                                                          return_indices=True)
        n = len(c["idx"])
        assert int(v[0]) == n, c["name"]
        assert idx[0, :n].tolist() == c["idx"], c["name"]
        assert k[0, :n].tolist() == c["cls"], c["name"]
        exp_b = c["boxes"][c["idx"]]
        if c["clip"]:
            exp_b = np.clip(exp_b, 0, 1)
        np.testing.assert_array_equal(b[0, :n], exp_b)
        np.testing.assert_array_equal(s[0, :n], c["scores"][c["idx"], c["cls"]])
        assert not b[0, n:].any() and not s[0, n:].any()       # zero padding rows


def test_preprocess_oracle_known_answers():
    """N4: uint8 -> [0,1] + bilinear resize (half-pixel centres).  Hand-checked cases: identity
    size, 2x upsampling of a ramp (edge clamping), 2x downsampling = mean of 2x2 blocks."""
    a = np.arange(12, dtype=np.uint8).reshape(2, 2, 3) * 20
    np.testing.assert_array_equal(bo.preprocess_image(a, 2, 2), a.astype(np.float32) * np.float32(1 / 255))
    ramp = np.array([[[0], [100]]], np.uint8)                       # 1 x 2 image
    up = bo.preprocess_image(ramp, 1, 4)[0, :, 0] * 255
    np.testing.assert_allclose(up, [0, 25, 75, 100], rtol=1e-5, atol=1e-4)     # src = -0.25, 0.25, 0.75, 1.25
    img = np.random.default_rng(0).integers(0, 256, (6, 8, 3), dtype=np.uint8)
    down = bo.preprocess_image(img, 3, 4)
    ref = img.astype(np.float64).reshape(3, 2, 4, 2, 3).mean((1, 3)) / 255
    np.testing.assert_allclose(down, ref, atol=1e-6)
    assert bo.preprocess_image(img[None], 5, 7).shape == (1, 5, 7, 3)
