"""N1 forward half: the HIP loss kernel behind ``ssd_loss.CustomLoss`` against the
NumPy restatement of reference ssd_loss.py:8-65 (oracle/loss_oracle.py) and, for the gradients,
against torch-CPU autograd of the same graph.  Bars: positives / hard-negative selection
(final_mask) bit-exact given the same per-anchor losses; per-anchor CE within 2e-6 (device logf
vs libm, last ulp); loss values within 1e-5 relative (summation order); gradients 1e-6 abs."""
import numpy as np
import pytest

import helpers
from oracle import loss_oracle as lo


def _case(B, N, L, seed, pos_frac=0.02, sharp=3.0):
    """Targets from the real target-assignment oracle statistics: a few positives per image with
    non-zero deltas and a one-hot label, background elsewhere; predictions = softmax(logits)."""
    rng = np.random.default_rng(seed)
    yl = np.zeros((B, N, L), np.float32)
    yd = np.zeros((B, N, 4), np.float32)
    for b in range(B):
        npos = int(rng.integers(0, max(2, int(pos_frac * N) * 2)))
        if b == 0:
            npos = 0                                   # an image without positives: divide by 1, no negatives
        idx = rng.choice(N, npos, replace=False)
        yl[b, :, 0] = 1
        yl[b, idx, 0] = 0
        yl[b, idx, rng.integers(1, L, npos)] = 1
        yd[b, idx] = rng.standard_normal((npos, 4)).astype(np.float32) * 2
    pd = (rng.standard_normal((B, N, 4)) * 1.5).astype(np.float32)
    z = (rng.standard_normal((B, N, L)) * sharp).astype(np.float32)
    z[..., 0] += 2.0
    e = np.exp(z - z.max(-1, keepdims=True))
    pp = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    return yd, yl, pd, z, pp


def test_oracle_loss_hand_values():
    """Known answers worked by hand: Huber branches, positive normalisation, 3:1 mining."""
    yd = np.zeros((1, 4, 4), np.float32)
    yd[0, 1] = [0.5, 0, 0, 0]
    yd[0, 2] = [0, 0, 2.0, 0]
    pd = np.zeros((1, 4, 4), np.float32)
    pd[0, 1] = [1.0, 0.25, 0, 0]          # errors .5,.25 -> .125 + .03125
    pd[0, 2] = [0, 0, -1.0, 3.0]          # errors -3, 3  -> 2.5 + 2.5
    pd[0, 3] = [9, 9, 9, 9]               # not a positive: ignored
    np.testing.assert_allclose(lo.loc_loss_fn(yd, pd, 2.0), [2.0 * (0.15625 + 5.0) / 2], rtol=1e-6)
    # conf: N = 6, one positive (class 2), total_neg = 3
    yl = np.zeros((1, 6, 3), np.float32)
    yl[0, :, 0] = 1
    yl[0, 0] = [0, 0, 1]
    pp = np.array([[[.2, .3, .5], [.9, .05, .05], [.5, .25, .25], [.5, .3, .2], [.99, .005, .005], [.6, .2, .2]]], np.float32)
    out, ce, fm = lo.conf_loss_fn(yl, pp, 3.0, return_parts=True)
    np.testing.assert_allclose(ce[0], -np.log([.5, .9, .5, .5, .99, .6]), rtol=3e-6)
    # negatives by descending loss: anchors 2 and 3 tie at -log(.5) (lower index first), then 5
    np.testing.assert_array_equal(fm[0], [1, 0, 1, 1, 0, 1])
    np.testing.assert_allclose(out, [(-np.log(.5) * 3 - np.log(.6)) / 1], rtol=1e-6)
    # more negatives requested than background anchors exist: positives rank too -> final_mask 2
    out2, _, fm2 = lo.conf_loss_fn(yl, pp, 6.0, return_parts=True)
    np.testing.assert_array_equal(fm2[0], [2, 1, 1, 1, 1, 1])
    # probabilities below the clip floor
    pz = np.array([[[1.0, 0.0, 0.0]]], np.float32)
    yz = np.array([[[0.0, 1.0, 0.0]]], np.float32)
    np.testing.assert_allclose(lo.cross_entropy(yz, pz), [[-np.log(np.float32(1e-7))]], rtol=1e-6)


def test_oracle_numpy_vs_torch_autograd_graph():
    yd, yl, pd, z, pp = _case(3, 200, 7, seed=1)
    loc, conf, probs, gd, gz = lo.torch_loss_and_grads(yd, yl, pd, z)
    np.testing.assert_allclose(probs, pp, atol=1e-6)
    np.testing.assert_allclose(loc, lo.loc_loss_fn(yd, pd), rtol=1e-5)
    np.testing.assert_allclose(conf, lo.conf_loss_fn(yl, probs), rtol=1e-5)
    assert np.abs(gd).max() > 0 and np.abs(gz).max() > 0
    # gradient lives only on selected anchors
    _, _, fm = lo.conf_loss_fn(yl, probs, return_parts=True)
    assert not gz[fm == 0].any() and not gd[~np.any(yd != 0, -1)].any()


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,L,seed", [(32, 2268, 21, 11), (32, 8732, 21, 12), (4, 24564, 21, 13), (2, 50, 3, 14),
                                        (3, 1, 21, 15), (5, 1025, 91, 16)])
def test_loss_kernel_vs_oracle(B, N, L, seed):
    import torch
    import ssd_hip as h
    from ssd_loss import CustomLoss
    yd, yl, pd, z, pp = _case(B, N, L, seed)
    cl = CustomLoss(3, 1)
    loc = cl.loc_loss_fn(yd, pd).cpu().numpy()
    conf = cl.conf_loss_fn(yl, pp).cpu().numpy()
    ce = cl.last_cross_entropy.cpu().numpy()
    fm = cl.last_final_mask.cpu().numpy()
    np.testing.assert_allclose(loc, lo.loc_loss_fn(yd, pd), rtol=1e-5, atol=1e-7)
    rce = lo.cross_entropy(yl, pp)
    assert np.abs(ce - rce).max() <= 2e-6 * max(1.0, float(rce.max()))
    # selection: bit-exact given the device's own per-anchor losses (ranks near-equal losses identically)
    rconf, _, rfm = lo.conf_loss_fn(yl, pp, 3.0, return_parts=True, ce=ce)
    np.testing.assert_array_equal(fm, rfm)
    np.testing.assert_allclose(conf, rconf, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(conf, lo.conf_loss_fn(yl, pp), rtol=1e-4, atol=1e-6)     # ... and end to end
    assert fm[0].sum() == 0 and conf[0] == 0 and loc[0] == 0       # image without positives
    npos = (yl[..., 1:] != 0).any(-1).sum(1)
    assert ((fm.sum(1) > 0) == (npos > 0)).all()
    # other ratio / alpha, both terms in one call with gradients
    cl2 = CustomLoss(1.5, 2.0)
    loc2, conf2, gd, gz = cl2.loss_and_grads(yd, yl, pd, pp)
    np.testing.assert_allclose(loc2.cpu().numpy(), lo.loc_loss_fn(yd, pd, 2.0), rtol=1e-5, atol=1e-7)
    rloc, rconf2, _, rgd, rgz = lo.torch_loss_and_grads(yd, yl, pd, z, 1.5, 2.0)
    np.testing.assert_allclose(conf2.cpu().numpy(), rconf2, rtol=1e-4, atol=1e-6)
    assert np.abs(gd.cpu().numpy() - rgd).max() <= 1e-6
    assert np.abs(gz.cpu().numpy() - rgz).max() <= 1e-6
    assert np.abs(rgz).max() > 1e-5


@pytest.mark.gpu
def test_loss_on_real_targets_and_errors():
    """Targets from calculate_actual_outputs (GPU matcher) at the C4 per-GPU shape."""
    from ssd_loss import CustomLoss
    from utils import bbox_utils, train_utils
    hp = helpers.hyper_params("mobilenet_v2")
    priors = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
    gt, gl = helpers.gt_inputs(32, G=16, seed=3)
    yd, yl = train_utils.calculate_actual_outputs(priors, gt, gl, hp)
    rng = np.random.default_rng(5)
    pd = (rng.standard_normal((32, 2268, 4)) * 0.5).astype(np.float32)
    z = rng.standard_normal((32, 2268, 21)).astype(np.float32)
    e = np.exp(z - z.max(-1, keepdims=True))
    pp = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    cl = CustomLoss(hp["neg_pos_ratio"], hp["loc_loss_alpha"])
    loc, conf = cl.loc_loss_fn(yd, pd).cpu().numpy(), cl.conf_loss_fn(yl, pp).cpu().numpy()
    ydn, yln = yd.cpu().numpy(), yl.cpu().numpy()
    np.testing.assert_allclose(loc, lo.loc_loss_fn(ydn, pd), rtol=1e-5)
    rconf, _, rfm = lo.conf_loss_fn(yln, pp, 3.0, return_parts=True, ce=cl.last_cross_entropy.cpu().numpy())
    np.testing.assert_array_equal(cl.last_final_mask.cpu().numpy(), rfm)
    np.testing.assert_allclose(conf, rconf, rtol=1e-5)
    pos = (yln[..., 1:] != 0).any(-1).sum(1)
    assert pos.max() > 100 and (pos * 4 < 2268).all()
    np.testing.assert_array_equal(rfm.sum(1), pos * 4)              # pos + 3 * pos hard negatives
    with pytest.raises(ValueError):
        cl.loc_loss_fn(ydn, pd[:, :10])
    with pytest.raises(ValueError):
        cl.conf_loss_fn(yln, pp[..., :20])
