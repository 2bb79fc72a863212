"""CPU tests of the network oracle: spec totals vs SURVEY.md Appendix C, TF padding rules,
and the NumPy conv restatement vs an independent implementation (torch CPU / oneDNN)."""
import numpy as np
import pytest

import helpers
from oracle import net_oracle as no
from oracle import torch_cpu_graph as tg


def test_spec_totals_match_survey():
    hp = helpers.hyper_params("mobilenet_v2")
    assert abs(no.count_macs("mobilenet_v2", hp) / 1e6 - 1013.1) < 0.1
    assert len(no.param_specs("mobilenet_v2", hp)) == 300
    assert abs(sum(int(np.prod(s)) for _, s in no.param_specs("mobilenet_v2", hp)) / 1e6 - 8.53) < 0.01
    hv = helpers.hyper_params("vgg16")
    assert abs(no.count_macs("vgg16", hv) / 1e6 - 31373.5) < 0.1
    assert abs(sum(int(np.prod(s)) for _, s in no.param_specs("vgg16", hv)) / 1e6 - 26.28) < 0.01


def test_tf_padding_rules():
    # SURVEY.md section 7 "hard parts": extras 10->5 (0,1), 5->3 (1,1), 3->2 (1,1), 2->1 (0,1)
    assert no.same_pads(10, 3, 2) == (5, 0, 1)
    assert no.same_pads(5, 3, 2) == (3, 1, 1)
    assert no.same_pads(3, 3, 2) == (2, 1, 1)
    assert no.same_pads(2, 3, 2) == (1, 0, 1)
    assert no.same_pads(19, 3, 2) == (10, 1, 1)
    assert no.same_pads(19, 3, 1, 6) == (19, 6, 6)      # conv6 dilation 6
    assert no.same_pads(300, 2, 2) == (150, 0, 0) and no.same_pads(75, 2, 2) == (38, 0, 1)
    assert no.correct_pad(300) == (0, 1) and no.correct_pad(75) == (1, 1) and no.correct_pad(19) == (1, 1)


@pytest.mark.parametrize("shape,k,stride,dil,padding", [
    ((2, 19, 19, 64), 3, 1, 1, "same"), ((1, 10, 10, 32), 3, 2, 1, "same"), ((1, 19, 19, 16), 3, 1, 6, "same"),
    ((2, 5, 5, 24), 3, 1, 1, "valid"), ((1, 33, 33, 3), 3, 2, 1, (0, 1, 0, 1)), ((2, 7, 7, 40), 1, 1, 1, "valid")])
def test_numpy_conv_vs_torch(shape, k, stride, dil, padding):
    rng = np.random.default_rng(1)
    x = rng.standard_normal(shape).astype(np.float32)
    w = rng.standard_normal((k, k, shape[3], 20)).astype(np.float32)
    b = rng.standard_normal(20).astype(np.float32)
    a = no.conv2d(x, w, b, stride, dil, padding)
    t = tg.TorchOps.output(tg.TorchOps.conv2d(tg.TorchOps.input(x), w, b, stride, dil, padding))
    np.testing.assert_allclose(a, t, atol=2e-4, rtol=1e-4)
    wd = rng.standard_normal((3, 3, shape[3], 1)).astype(np.float32)
    if k == 3 and dil == 1:
        a = no.depthwise_conv2d(x, wd, stride, padding)
        t = tg.TorchOps.output(tg.TorchOps.depthwise_conv2d(tg.TorchOps.input(x), wd, stride, padding))
        np.testing.assert_allclose(a, t, atol=1e-4, rtol=1e-4)


def test_pool_l2norm_softmax_vs_torch():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((1, 19, 19, 8)).astype(np.float32)
    for k, s in ((2, 2), (3, 1)):
        np.testing.assert_array_equal(no.max_pool(x, k, s),
                                      tg.TorchOps.output(tg.TorchOps.max_pool(tg.TorchOps.input(x), k, s)))
    g = rng.uniform(10, 30, 8).astype(np.float32)
    np.testing.assert_allclose(no.l2_normalize_scale(x, g),
                               tg.TorchOps.output(tg.TorchOps.l2_normalize_scale(tg.TorchOps.input(x), g)), rtol=2e-6)
    lg = rng.standard_normal((5, 7, 21)).astype(np.float32)
    np.testing.assert_allclose(no.softmax(lg), tg.TorchOps.softmax(tg._t(lg)).numpy(), atol=1e-7)


def test_full_graphs_numpy_vs_torch():
    """Whole MobileNetV2-SSD300 graph: NumPy restatement vs torch-CPU, same weights."""
    hp = helpers.hyper_params("mobilenet_v2")
    w = helpers.synthetic_weights("mobilenet_v2", hp)
    x = helpers.images(1, 300, seed=0)
    d, p = no.forward("mobilenet_v2", hp, w, x)
    td, tp = tg.forward("mobilenet_v2", hp, w, x)
    assert d.shape == (1, 2268, 4) and p.shape == (1, 2268, 21)
    assert np.abs(p - tp).max() < 1e-4 and np.abs(d - td).max() < 1e-4 * max(1, np.abs(d).max())
    frac = float(((p.argmax(-1) != 0) & (p.max(-1) > 0.5)).mean())
    assert 0.02 < frac < 0.10          # calibrated ~5 % non-background anchors above threshold
