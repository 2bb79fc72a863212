"""External anchors for everything that runs inside TensorFlow in the reference.

1. ``tests/golden/tf_*.npz`` -- outputs of the REAL reference executed on TensorFlow, written by
   ``tools/make_tf_golden.py`` on a machine that has TF (this container and the GPU boxes do not).  While a file
   is absent its tests SKIP (they do not pass): the parity of that piece stays "pinned by the restatement only".
   When present, the CPU oracle (always) and the HIP path (``-m gpu``) are held to the reference's own numbers.
2. [3P] published known answers that need no TensorFlow here: the worked examples of the Keras API documentation
   for ``tf.keras.losses.Huber`` (0.155) and ``CategoricalCrossentropy`` (1.177), and the TF2 half-pixel
   bilinear resize of a 2x2 image to 4x4 -- through the oracle (CPU) and through the HIP kernels (GPU).
"""
import os

import numpy as np
import pytest

import helpers
from oracle import bbox_oracle as bo
from oracle import loss_oracle as lo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _tf(name):
    p = os.path.join(GOLD, name)
    if not os.path.exists(p):
        pytest.skip("%s absent: generate it where TensorFlow exists (tools/make_tf_golden.py --reference <tf-ssd checkout>)" % name)
    return np.load(p)


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


# ------------------------------------------------------------------ [3P] published known answers (no TF needed)
# Keras docs, tf.keras.losses.Huber: y_true = [[0, 1], [0, 0]], y_pred = [[0.6, 0.4], [0.4, 0.6]] -> 0.155
# (element losses 0.18, 0.18, 0.08, 0.18; mean).  As ONE positive anchor with four coordinates the SSD
# localisation loss (ssd_loss.py:18-31: Huber summed over the coordinates of the positives / positives) is the
# sum of the same four numbers = 4 x 0.155.
_HUBER_TRUE = np.array([[[0, 1, 0, 0]]], np.float32)
_HUBER_PRED = np.array([[[0.6, 0.4, 0.4, 0.6]]], np.float32)
# Keras docs, tf.keras.losses.CategoricalCrossentropy: y_true = [[0, 1, 0], [0, 0, 1]],
# y_pred = [[0.05, 0.95, 0], [0.1, 0.8, 0.1]] -> 1.177 (= (-ln 0.95 - ln 0.1) / 2).  Two positive anchors plus six
# perfectly predicted background anchors (the 3 x 2 hard negatives the mining wants -- ssd_loss.py:52-58 would
# otherwise pick the positives a second time -- each adding ~1e-7): the SSD confidence loss (ssd_loss.py:45-63) is
# the cross-entropy sum / positives.
_CCE_TRUE = np.array([[[0, 1, 0], [0, 0, 1]] + [[1, 0, 0]] * 6], np.float32)
_CCE_PRED = np.array([[[0.05, 0.95, 0.0], [0.1, 0.8, 0.1]] + [[1, 0, 0]] * 6], np.float32)
# TF2 ``tf.image.resize`` (bilinear, half-pixel centres, no antialias) of [[1, 2], [3, 4]] to 4 x 4
_RESIZE_4x4 = np.array([[1, 1.25, 1.75, 2], [1.5, 1.75, 2.25, 2.5], [2.5, 2.75, 3.25, 3.5], [3, 3.25, 3.75, 4]], np.float64)


def test_published_loss_examples_oracle():
    assert abs(float(lo.loc_loss_fn(_HUBER_TRUE, _HUBER_PRED)[0]) / 4 - 0.155) < 5e-4
    assert abs(float(lo.conf_loss_fn(_CCE_TRUE, _CCE_PRED, 3.0)[0]) - 1.177) < 5e-4


def test_published_resize_example_oracle():
    img = np.repeat(np.array([[1, 2], [3, 4]], np.uint8)[:, :, None], 3, axis=2)
    out = bo.preprocess_image(img, 4, 4).astype(np.float64) * 255.0
    for c in range(3):
        np.testing.assert_allclose(out[:, :, c], _RESIZE_4x4, rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_published_examples_hip():
    from ssd_loss import CustomLoss
    from utils import data_utils
    cl = CustomLoss(3, 1)
    assert abs(float(_np(cl.loc_loss_fn(_HUBER_TRUE, _HUBER_PRED))[0]) / 4 - 0.155) < 5e-4
    assert abs(float(_np(cl.conf_loss_fn(_CCE_TRUE, _CCE_PRED))[0]) - 1.177) < 5e-4
    img = np.repeat(np.array([[1, 2], [3, 4]], np.uint8)[:, :, None], 3, axis=2)
    out = _np(data_utils.preprocess_batch(img[None], 4, 4))[0].astype(np.float64) * 255.0
    for c in range(3):
        np.testing.assert_allclose(out[:, :, c], _RESIZE_4x4, rtol=0, atol=1e-5)


# ------------------------------------------------------------------ reference-executed fixtures (skip while absent)
def test_tf_priors_vs_oracle():
    z = _tf("tf_priors.npz")
    for bb in ("mobilenet_v2", "vgg16"):
        np.testing.assert_array_equal(bo.generate_prior_boxes(helpers.FMAPS[bb], helpers.ASPECT_RATIOS), z[bb])


def _decoder_cases():
    z = _tf("tf_decode_nms.npz")
    src = np.load(os.path.join(GOLD, "decode_nms.npz"))
    for name in ("rand", "none", "degenerate", "ties"):
        yield name, src[name + "_deltas"], src[name + "_probs"], z[name + "_boxes"], z[name + "_labels"], z[name + "_scores"]


def _check_decode(name, b, l, s, rb, rl, rs):
    if name == "ties":      # TF leaves the order of equal scores unspecified: compare as sets of (label, score)
        for i in range(b.shape[0]):
            assert sorted(zip(l[i].tolist(), s[i].tolist())) == sorted(zip(rl[i].tolist(), rs[i].tolist()))
        return
    np.testing.assert_array_equal(l, rl)
    assert np.abs(s - rs).max() <= 1e-6 and np.abs(b - rb).max() <= 1e-4


def test_tf_decoder_vs_oracle():
    pri = bo.generate_prior_boxes(helpers.FMAPS["mobilenet_v2"], helpers.ASPECT_RATIOS)
    for name, d, p, rb, rl, rs in _decoder_cases():
        b, l, s, _v = bo.ssd_decode(pri, helpers.VARIANCES, d, p)
        _check_decode(name, b, l, s, rb, rl, rs)


@pytest.mark.gpu
def test_tf_decoder_vs_hip():
    from models.decoder import SSDDecoder
    from utils import bbox_utils
    pri = bbox_utils.generate_prior_boxes(helpers.FMAPS["mobilenet_v2"], helpers.ASPECT_RATIOS)
    dec = SSDDecoder(pri, helpers.VARIANCES)
    for name, d, p, rb, rl, rs in _decoder_cases():
        b, l, s = (_np(t) for t in dec([d, p]))
        _check_decode(name, b, l, s, rb, rl, rs)


def test_tf_match_vs_oracle():
    z = _tf("tf_match.npz")
    m = np.load(os.path.join(GOLD, "match.npz"))
    pri = bo.generate_prior_boxes(helpers.FMAPS["mobilenet_v2"], helpers.ASPECT_RATIOS)
    dl, oh, lab, _mi = bo.calculate_actual_outputs(pri, m["gt"], m["gl"], helpers.hyper_params(), return_indices=True)
    np.testing.assert_array_equal(oh, z["labels_one_hot"])
    assert np.abs(dl - z["deltas"]).max() <= 1e-5
    np.testing.assert_allclose(bo.generate_iou_map(pri, m["gt"]), z["iou"], rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_tf_match_vs_hip():
    from utils import bbox_utils, train_utils
    z = _tf("tf_match.npz")
    m = np.load(os.path.join(GOLD, "match.npz"))
    pri = bbox_utils.generate_prior_boxes(helpers.FMAPS["mobilenet_v2"], helpers.ASPECT_RATIOS)
    dl, oh = train_utils.calculate_actual_outputs(pri, m["gt"], m["gl"], helpers.hyper_params())
    np.testing.assert_array_equal(_np(oh), z["labels_one_hot"])
    assert np.abs(_np(dl) - z["deltas"]).max() <= 1e-5


def _loss_cases():
    from test_loss import _case
    z = _tf("tf_loss.npz")
    for B, N, L, seed in ((4, 2268, 21, 11), (2, 8732, 21, 12), (2, 50, 3, 14)):
        yd, yl, pd, _z, pp = _case(B, N, L, seed)
        key = "b%d_n%d_l%d_s%d" % (B, N, L, seed)
        yield yd, yl, pd, pp, z[key + "_loc"], z[key + "_conf"]


def test_tf_loss_vs_oracle():
    for yd, yl, pd, pp, rloc, rconf in _loss_cases():
        np.testing.assert_allclose(lo.loc_loss_fn(yd, pd), rloc, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(lo.conf_loss_fn(yl, pp, 3.0), rconf, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_tf_loss_vs_hip():
    from ssd_loss import CustomLoss
    cl = CustomLoss(3, 1)
    for yd, yl, pd, pp, rloc, rconf in _loss_cases():
        np.testing.assert_allclose(_np(cl.loc_loss_fn(yd, pd)), rloc, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(_np(cl.conf_loss_fn(yl, pp)), rconf, rtol=1e-5, atol=1e-6)


def test_tf_published_file_matches_the_documented_values():
    z = _tf("tf_published.npz")
    assert abs(float(z["huber"]) - 0.155) < 5e-4 and abs(float(z["cce"]) - 1.177) < 5e-4
    np.testing.assert_allclose(bo.preprocess_image(z["resize_in"], 30, 30), z["resize_out"], rtol=0, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("backbone", ["mobilenet_v2", "vgg16"])
def test_tf_network_vs_hip(backbone):
    """The whole graph on TensorFlow (Keras conv / BN / pooling numerics, [3P] keras-applications MobileNetV2) vs
    the HIP forward, same seeded weights and images: the contract's 1e-4 on probabilities and variance-scaled deltas."""
    z = _tf("tf_net_%s.npz" % backbone)
    if backbone == "mobilenet_v2":
        from models.ssd_mobilenet_v2 import get_model
    else:
        from models.ssd_vgg16 import get_model
    hp = helpers.hyper_params(backbone)
    m = get_model(hp)
    m.set_weights(helpers.synthetic_weights(backbone, hp))
    d, p = m(helpers.images(2, 300, seed=0))
    assert np.abs(_np(p) - z["probs"]).max() <= 1e-4
    assert np.abs((_np(d) - z["deltas"]) * np.asarray(helpers.VARIANCES, np.float32)).max() <= 1e-4
