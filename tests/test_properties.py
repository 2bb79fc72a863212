"""Property-based tests (hypothesis; SURVEY.md 4 iv) of the decode / NMS / matching path: random
shapes, thresholds and degenerate inputs.  CPU: the two independent oracle restatements (NumPy
loops vs plain C) must agree and satisfy the CombinedNMS contract.  GPU: the HIP path must agree
with the C oracle bit for bit on indices / labels / scores."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import helpers
from oracle import bbox_oracle as bo
from oracle import c_oracle as co

_shapes = st.tuples(st.integers(1, 3), st.integers(1, 180), st.integers(2, 9), st.integers(0, 2 ** 31 - 1),
                    st.sampled_from([0.3, 0.5, 0.7]), st.sampled_from([0.05, 0.3, 0.5]), st.integers(1, 30))


def _case(B, N, L, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(0.0, 1.0, (N, 2)); sz = rng.uniform(0.0, 0.5, (N, 2))
    pri = np.clip(np.concatenate([c - sz / 2, c + sz / 2], -1), 0, 1).astype(np.float32)
    if N > 3:
        pri[rng.integers(0, N)] = pri[rng.integers(0, N)]          # duplicate prior -> IoU 1 pairs
        pri[rng.integers(0, N), 2:] = pri[rng.integers(0, N), :2]  # degenerate prior
    d = (rng.standard_normal((B, N, 4)) * rng.choice([0.0, 0.5, 3.0])).astype(np.float32)
    p = rng.random((B, N, L)).astype(np.float32) ** rng.choice([1, 3])
    p /= p.sum(-1, keepdims=True)
    if rng.random() < 0.3:
        p[:, ::2] = p[:, :1]                                          # tied scores
    return pri, d, p.astype(np.float32)


def _contract(b, l, s, v, T, thr):
    for i in range(b.shape[0]):
        n = int(v[i])
        assert n <= T and (s[i, :n] > thr).all() and (np.diff(s[i, :n]) <= 0).all()
        assert not s[i, n:].any() and not b[i, n:].any() and not l[i, n:].any()
        assert b[i].min() >= 0 and b[i].max() <= 1 and (l[i, :n] >= 0).all()     # column 0 is an ordinary class below 0.5 (models/decoder.py:45)


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(_shapes)
def test_decoder_numpy_vs_c_oracle(args):
    B, N, L, seed, iou, thr, T = args
    pri, d, p = _case(B, N, L, seed)
    rb, rl, rs, rv, ri = co.decode_nms(d, p, pri, helpers.VARIANCES, max_per_class=T, max_total=T, iou_thr=iou, score_thr=thr)
    nb, nl, ns = bo.ssd_decode(pri, helpers.VARIANCES, d, p, max_total_size=T, score_threshold=thr, iou_threshold=iou)[:3]
    np.testing.assert_array_equal(nl, rl)
    np.testing.assert_array_equal(ns, rs)
    np.testing.assert_allclose(nb, rb, atol=2e-6)
    _contract(rb, rl, rs, rv, T, thr)


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(st.tuples(st.integers(1, 3), st.integers(1, 120), st.integers(1, 12), st.integers(0, 2 ** 31 - 1)))
def test_matching_numpy_vs_c_oracle(args):
    B, N, G, seed = args
    rng = np.random.default_rng(seed)
    pri, _, _ = _case(1, N, 2, seed)
    gt, gl = helpers.gt_inputs(B, G=G, L=21, seed=seed % 1000)
    hp = helpers.hyper_params("mobilenet_v2")
    dl, lab, midx = co.match_encode(pri, gt, gl, hp["variances"], 0.5)
    rd, roh, rlab, rmidx = bo.calculate_actual_outputs(pri, gt, gl, hp, return_indices=True)
    np.testing.assert_array_equal(midx, rmidx)
    np.testing.assert_array_equal(lab, rlab)
    np.testing.assert_allclose(dl, rd, atol=2e-6)
    assert (roh.sum(-1) == 1).all() and not rd[rlab == 0].any()


@pytest.mark.gpu
@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(_shapes)
def test_decoder_gpu_vs_c_oracle(args):
    from models.decoder import SSDDecoder
    B, N, L, seed, iou, thr, T = args
    pri, d, p = _case(B, N, L, seed)
    dec = SSDDecoder(pri, helpers.VARIANCES, max_total_size=T, score_threshold=thr)
    dec.iou_threshold = iou
    b, l, s = dec.call([d, p], return_indices=True)
    rb, rl, rs, rv, ri = co.decode_nms(d, p, pri, helpers.VARIANCES, max_per_class=T, max_total=T, iou_thr=iou, score_thr=thr)
    np.testing.assert_array_equal(dec.last_kept_indices.cpu().numpy(), ri)
    np.testing.assert_array_equal(dec.last_valid_detections.cpu().numpy(), rv)
    np.testing.assert_array_equal(l.cpu().numpy(), rl)
    np.testing.assert_array_equal(s.cpu().numpy(), rs)
    np.testing.assert_allclose(b.cpu().numpy(), rb, atol=1e-5)
